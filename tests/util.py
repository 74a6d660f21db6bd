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
        nz = ocnt > 0       # for an empty cell only start == end is specified
        assert np.array_equal(ecs[nz], ocs[nz]) and np.array_equal(ece[nz], oce[nz]), what + "cell ranges differ"
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


# ---- golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py) ----------------
import ctypes as _C
import glob as _glob
import os as _os

GOLDEN_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(_os.path.splitext(_os.path.basename(p))[0] for p in _glob.glob(_os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    fx = dict(np.load(_os.path.join(GOLDEN_DIR, name + ".npz")))
    p = abi.SphParams()
    raw = fx["params"].tobytes()
    assert len(raw) == _C.sizeof(abi.SphParams), "fixture was written with a different sph_params layout"
    _C.memmove(_C.byref(p), raw, len(raw))
    fx["params_struct"] = p
    return fx


def replay_golden(engine, fx):
    """Upload the fixture's input state, run ONE frame through the engine-like object."""
    p = fx["params_struct"]
    n = fx["state_in"].shape[0]
    engine.alloc_particles(n)
    engine.upload_particles(fx["state_in"])
    engine.resize_grid(p.gridResolutionX, p.gridResolutionY, p.gridResolutionZ, p.maxParticlesPerCell)
    engine.set_params(p)
    for k in range(3):
        engine.reset_counter(k)
    engine.step_frame(int(fx["sub_steps"]), float(fx["time_step"]))


def golden_neighbor_lists(fx):
    cnt, flat = fx["neighbor_counts"], fx["neighbor_flat"]
    off = np.r_[0, np.cumsum(cnt)]
    return [flat[off[i]:off[i + 1]] for i in range(len(cnt))]


def check_against_golden(engine, fx, tol, exact_fields=False, what=""):
    cap = int(fx["params_struct"].maxParticlesPerCell)
    cs, ce, ids, cop = engine.download_grid()[:4]
    assert np.array_equal(cop, fx["cell_of_particle"]), what + "cell hash differs"
    nz = fx["grid_counts_nonzero_cells"]
    counts = np.zeros(cs.shape[0], dtype=np.int32)
    counts[nz] = fx["grid_counts_nonzero"]
    overflow = int(counts.max(initial=0)) > cap
    if not overflow:
        assert np.array_equal(ce - cs, counts), what + "cell populations differ"
        assert np.array_equal(ids, fx["sorted_ids"]), what + "sorted ids differ"
    ncnt, nlst = engine.download_neighbors(512)
    gl = golden_neighbor_lists(fx)
    assert np.array_equal(ncnt, fx["neighbor_counts"]), what + "neighbour counts differ"
    for i, want in enumerate(gl):
        if not np.array_equal(nlst[i, :len(want)], want):
            raise AssertionError(what + f"neighbour list of particle {i} differs")
    got, want = engine.download_particles(), fx["state_out"]
    if exact_fields:
        assert np.array_equal(got, want), what + "state differs bitwise"
    else:
        assert_fields_close(got, want, tol, what)
    ctr = [engine.read_counter(k) for k in range(3)]
    assert ctr == fx["counters"].tolist(), what + f"counters {ctr} != {fx['counters'].tolist()}"
