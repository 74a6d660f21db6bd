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


# ---- engine and oracle driven as ONE engine (kernel-granular lock-step) ---------------------------
class Pair:
    """Looks like one engine to sph_amd.host.SPH and forwards every call to the engine AND the oracle, so that one
    host driver steps both through a frame kernel by kernel.  `after(kernel_id)` runs behind every dispatch: that is
    where tests compare mid-frame artefacts (the hit mask behind CS_DensityPressure) and re-synchronise the engine to
    the oracle between sub-steps."""
    grid_first = False

    def __init__(self, eng, ora, after=None):
        self.eng, self.ora, self.after = eng, ora, after

    @property
    def n(self):
        return self.eng.n

    def dispatch(self, kernel_id, count=None):
        self.eng.dispatch(kernel_id, count)
        self.ora.dispatch(kernel_id, count)
        if self.after:
            self.after(kernel_id)

    def download_particles(self):
        return self.ora.download_particles()

    def destroy(self):
        self.eng.destroy()
        self.ora.destroy()

    def __getattr__(self, name):
        if name.startswith("_") or not hasattr(self.ora, name):
            raise AttributeError(name)
        fe, fo = getattr(self.eng, name), getattr(self.ora, name)

        def both(*a, **k):
            r = fe(*a, **k)
            fo(*a, **k)
            return r
        return both


def assert_hit_lists_equal(eng, ora, max_nb=512, what="", expect_overflow=None):
    """The bit mask the force kernel is about to walk, decoded (sph_download_hit_lists), against the oracle's
    neighbour lists (compute:170-181 as the oracle restates it) -- bit-exact, in visiting order.  Returns the mask
    state; with a bin over its capacity the engine must have recorded NO mask (state 2: the capped scan path runs)."""
    state, ec, el = eng.download_hit_lists(max_nb)
    overflow = ora.read_counter(abi.CTR_BIN_OVERFLOW) > 0
    if expect_overflow is not None:
        assert overflow == expect_overflow, what + f"oracle overflow flag is {overflow}"
    if overflow:
        assert state == 2, what + f"a bin overflowed but the engine reports mask state {state}"
        return state
    assert state == 1, what + f"no fresh hit mask behind CS_DensityPressure (state {state})"
    oc, ol = ora.download_neighbors(max_nb)
    assert int(oc.max(initial=0)) <= max_nb
    have = ec >= 0                       # -1: the particle's rows need more mask words than a row holds (its wave scans)
    assert np.all(ec[~have] == -1)
    assert have.mean() > 0.9, what + f"only {have.mean():.2%} of the particles have a mask"
    assert np.array_equal(ec[have], oc[have]), what + "hit counts differ from the oracle's neighbour counts"
    assert np.array_equal(el[have], ol[have]), what + "hit lists differ from the oracle's neighbour lists"
    return state


def run_stepwise(builder, frames, tol=1e-5, max_nb=1024, want_overflow=None, min_masked=1, omp=False):
    """builder(engine) -> Scene.  Returns statistics of what was checked."""
    from oracle.ora import Oracle
    from sph_amd import abi
    eng, ora = abi.Engine(0), Oracle(omp=omp)
    stats = dict(masks=0, overflow_masks=0, substeps=0)
    where = [""]

    def after(kid):
        if kid == abi.K_DENSITY_PRESSURE:
            state = assert_hit_lists_equal(eng, ora, max_nb, where[0] + "behind CS_DensityPressure: ")
            stats["masks" if state == 1 else "overflow_masks"] += 1
        elif kid == abi.K_BOUND_OBS:
            # end of a sub-step: plain bar, then continue from the oracle's state (grid kept)
            state = ora.download_particles()
            assert_fields_close(eng.download_particles(), state, tol, where[0] + f"sub-step {stats['substeps']}: ")
            eng.upload_particles_in_place(state)
            stats["substeps"] += 1

    pair = Pair(eng, ora, after)
    sc = builder(pair)
    sc.sph.use_step_frame = False
    sc.start()
    for f in range(frames):
        where[0] = f"frame {f}: "
        eng.upload_particles(ora.download_particles())
        sc.frame()
        cap = sc.sph.maxParticlesPerCell
        assert_grid_equal(eng.download_grid(), ora.download_grid(with_counts=True), cap, where[0])
        assert_neighbors_equal(eng, ora, max_nb, where[0])
        assert_fields_close(eng.download_particles(), ora.download_particles(), tol, where[0])
        for ctr in range(3):
            assert eng.read_counter(ctr) == ora.read_counter(ctr), where[0] + f"counter {ctr}"
    pair.destroy()
    assert stats["masks"] + stats["overflow_masks"] == stats["substeps"] > 0
    if want_overflow is True:
        assert stats["overflow_masks"] > 0, "the case was meant to overflow a bin"
    if want_overflow is False:
        assert stats["overflow_masks"] == 0
    assert stats["masks"] >= min_masked
    return stats
