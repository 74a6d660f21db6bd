"""Shared helpers of the parity tests."""
import numpy as np

from sph_amd import abi

POS, VEL, ACC = slice(0, 3), slice(3, 6), slice(6, 9)
DENS, PRES, RHO0, VISC, MASS = 9, 10, 11, 12, 13
COL = slice(14, 18)


def field_scale(a):
    return max(float(np.max(np.abs(a))), 1e-30)


def assert_fields_close(got, want, tol=1e-5, what=""):
    """|got - want| <= tol * max|want| per field group (relative to FIELD SCALE: the Tait
    pressure cancels catastrophically near rho0, so element-wise relative error is meaningless)."""
    for name, sl in (("pos", POS), ("vel", VEL), ("acc", ACC), ("density", DENS), ("pressure", PRES)):
        g, w = got[:, sl], want[:, sl]
        err = float(np.max(np.abs(g.astype(np.float64) - w.astype(np.float64)))) if g.size else 0.0
        scale = field_scale(w)
        assert err <= tol * scale, f"{what}{name}: max abs err {err:.3e} > {tol:g} * scale {scale:.3e}"
    for name, sl in (("rho0", RHO0), ("visc", VISC), ("mass", MASS), ("color", COL)):
        assert np.array_equal(got[:, sl], want[:, sl]), f"{what}{name} not bit-identical"


def assert_grid_equal(eng_grid, ora_grid, cap, what=""):
    """Hash, sorted order and ranges bit-exact.  ora_grid = (cs, ce, ids, cop, counts)."""
    ecs, ece, eids, ecop = eng_grid
    ocs, oce, oids, ocop, ocnt = ora_grid
    assert np.array_equal(ecop, ocop), what + "cell_of_particle differs"
    assert np.array_equal(ece - ecs, ocnt), what + "cell populations differ from GridCounts"
    if int(ocnt.max(initial=0)) <= cap:
        assert np.array_equal(ecs, ocs) and np.array_equal(ece, oce), what + "cell ranges differ"
        assert np.array_equal(eids, oids), what + "sorted id sequence differs"
    else:   # overflow: the oracle stored only the first cap ids of each bin
        nz = np.nonzero(ocnt)[0]
        for c in nz:
            k = min(int(ocnt[c]), cap)
            assert np.array_equal(eids[ecs[c]:ecs[c] + k], oids[ocs[c]:ocs[c] + k]), what + f"bin {c} differs"


def assert_neighbors_equal(eng, ora, max_nb=512, what=""):
    ec, el = eng.download_neighbors(max_nb)
    oc, ol = ora.download_neighbors(max_nb)
    assert np.array_equal(ec, oc), what + "neighbour counts differ"
    assert int(oc.max(initial=0)) <= max_nb
    assert np.array_equal(el, ol), what + "neighbour lists differ"
    return oc
