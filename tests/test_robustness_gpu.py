"""States a healthy run never reaches, which the engine must still survive the way the reference's kernels would:
crowded boundary cells (every escaped particle is clamped into one, compute:126), non-finite particles, parameter
values that break an internal encoding (negative Tait pressure), kernels dispatched in an unusual order.  -m gpu."""
import time

import numpy as np
import pytest

from sph_amd import abi, scenes
from tests.util import assert_fields_close, assert_grid_equal, assert_neighbors_equal

pytestmark = pytest.mark.gpu


def _pair(builder):
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle(omp=True)
    se, so = builder(eng), builder(ora)
    return se, so, eng, ora


def test_fifty_thousand_escaped_particles_crowd_the_boundary_cells():
    """50,000 particles uploaded OUTSIDE the grid: half of them clamp into ONE boundary cell, the rest spread over a
    face.  The grid build (rank inside a cell = number of smaller ids) must finish in bounded time with the sorted ids
    bit-exact against the oracle's bins (first `cap` ids of every bin: rule O2), and the frame that follows must agree
    with the oracle like any other."""
    n = 50000
    se, so, eng, ora = _pair(lambda e: scenes.config1(e, n=n))
    se.sph.use_step_frame = True
    se.start()
    so.start()
    state = ora.download_particles()
    rng = np.random.default_rng(7)
    # config 1: AABB x in [-5, 5], y in [0, 10], z in [-5, 5]; cell 0.25
    state[: n // 2, 0] = 6.0 + 3.0 * rng.random(n // 2).astype(np.float32)        # beyond the +x face ...
    state[: n // 2, 1] = 2.0 + 0.2 * rng.random(n // 2).astype(np.float32)        # ... all into the same (y, z) cell
    state[: n // 2, 2] = 0.0 + 0.2 * rng.random(n // 2).astype(np.float32)
    state[n // 2:, 0] = -7.0 - rng.random(n - n // 2).astype(np.float32)          # beyond the -x face, spread over it
    state[n // 2:, 1] = 10.0 * rng.random(n - n // 2).astype(np.float32)
    state[n // 2:, 2] = -5.0 + 10.0 * rng.random(n - n // 2).astype(np.float32)
    perm = rng.permutation(n)                                                      # ids and places unrelated
    state[:, 0:3] = state[perm, 0:3]
    ora.upload_particles(state)
    eng.upload_particles(state)
    t0 = time.perf_counter()
    se.frame()
    eng.synchronize()
    dt = time.perf_counter() - t0
    so.frame()
    cap = so.sph.maxParticlesPerCell
    og = ora.download_grid(with_counts=True)
    assert int(og[4].max()) >= n // 2 > cap
    assert_grid_equal(eng.download_grid(), og, cap)
    assert eng.read_counter(abi.CTR_OUT_OF_GRID) == ora.read_counter(abi.CTR_OUT_OF_GRID) == n
    assert eng.read_counter(abi.CTR_BIN_OVERFLOW) == ora.read_counter(abi.CTR_BIN_OVERFLOW) > 0
    assert_fields_close(eng.download_particles(), ora.download_particles(), 1e-5)
    assert dt < 5.0, f"frame with crowded cells took {dt:.2f} s"
    eng.destroy()


def test_one_million_particles_in_one_cell_grid_build_is_bounded():
    """The worst case of the same state: EVERY particle in one bin (a diverged run).  Serial ranking was 10^12
    dependent loads here; the wave-cooperative ranking finishes, and the bin comes out in ascending id."""
    n = 1 << 20
    eng = abi.Engine(0)
    sc = scenes.config1(eng, n=n)
    sc.start()
    state = eng.download_particles()
    state[:, 0] = 50.0
    state[:, 1] = 3.1
    state[:, 2] = 0.1
    eng.upload_particles(state)
    t0 = time.perf_counter()
    eng.dispatch(abi.K_CLEAR_GRID, eng.cells)
    eng.dispatch(abi.K_BUILD_GRID, n)
    eng.synchronize()
    dt = time.perf_counter() - t0
    cs, ce, ids, cop = eng.download_grid()
    assert len(np.unique(cop)) == 1 and int((ce - cs).max()) == n
    assert np.array_equal(ids, np.arange(n, dtype=np.int32)), "a bin must list its ids in ascending order (rule O1)"
    assert dt < 20.0, f"grid build with 2^20 particles in one cell took {dt:.1f} s"
    print(f"2^20 particles in one cell: grid build {dt * 1e3:.0f} ms")
    eng.destroy()


def test_non_finite_particles_are_counted_not_fatal():
    se, so, eng, ora = _pair(lambda e: scenes.config1(e, n=4096))
    se.sph.use_step_frame = True
    se.start()
    state = eng.download_particles()
    clean = state.copy()
    state[10, 0] = np.nan
    state[20, 1] = np.nan
    state[30, 2] = np.nan
    state[40, 4] = np.inf
    eng.upload_particles(state)
    for _ in range(2):
        se.frame()
        assert eng.read_counter(abi.CTR_NAN) == 4
    out = eng.download_particles()
    bad = ~np.isfinite(out[:, 0:6]).all(axis=1)
    assert sorted(np.flatnonzero(bad).tolist()) == [10, 20, 30, 40], "non-finite values spread to other particles"
    eng.upload_particles(clean)
    se.frame()
    assert eng.read_counter(abi.CTR_NAN) == 0
    assert np.isfinite(eng.download_particles()).all()
    eng.destroy()


@pytest.mark.parametrize("step_frame", [False, True])
def test_negative_tait_pressure_switches_the_packed_records_off(step_frame):
    """gamma < 0 (Inspector-editable, SPH.cs:7) gives Tait pressures of -0.0; the packed neighbour records keep
    the phase id in the SIGN of density and pressure and must not be used then.  Compared with the oracle, which
    simply uses the values like the reference does."""
    def builder(e):
        sc = scenes.unity_scene(e, which=2, scale_count=0.02)
        sc.sph.gamma = -1.5
        return sc
    se, so, eng, ora = _pair(builder)
    se.sph.use_step_frame = step_frame
    se.start()
    so.start()
    for f in range(3):
        eng.upload_particles(ora.download_particles())
        se.frame()
        so.frame()
        want = ora.download_particles()
        # B = rho0 c^2 / gamma < 0 times ((rho/rho0)^gamma - 1) <= 0: pressures >= 0 again, but -0.0 wherever the density
        # sits on its floor -- a set sign bit, which the packed records would read as phase bit 1
        assert np.signbit(want[:, 10]).any(), "the case needs pressures with the sign bit set"
        assert_neighbors_equal(eng, ora, 512)
        assert_fields_close(eng.download_particles(), want, 1e-5, f"frame {f}: ")
    eng.destroy()


def test_full_step_between_density_and_force_is_seen_by_the_force_kernel():
    """A caller may dispatch any kernel at any time (SPH.DispatchCompute is public to the script).  CS_VvFullStep between
    CS_DensityPressure and CS_ForceXSPH changes the velocities the force kernel must read: records packed by the
    density kernel would be stale."""
    se, so, eng, ora = _pair(lambda e: scenes.unity_scene(e, which=4, scale_count=0.02))
    se.start()
    so.start()
    se.frame()
    so.frame()                                    # one ordinary frame: accelerations are non-zero now
    eng.upload_particles(ora.download_particles())
    n = eng.n
    order = [abi.K_CLEAR_GRID, abi.K_BUILD_GRID, abi.K_VV_HALF_STEP, abi.K_CLEAR, abi.K_DENSITY_PRESSURE,
             abi.K_VV_FULL_STEP, abi.K_VELOCITY_DAMPING, abi.K_FORCE_XSPH, abi.K_BOUND_OBS]
    for e in (eng, ora):
        e.set_phase_table(se.sph._phase_rows())
        e.apply_phase_table()
        e.set_delta_time(0.02)
        for k in order:
            e.dispatch(k, e.cells if k == abi.K_CLEAR_GRID else n)
    assert_fields_close(eng.download_particles(), ora.download_particles(), 1e-5)
    eng.destroy()


def test_extreme_positions_hash_and_search_like_the_oracle():
    """Positions far outside any grid: +-1e9, +-3e38, +-inf, NaN.  float -> int conversion saturates and maps NaN to 0
    (v_cvt_i32_f32, D3D ftoi, the oracle's ftoi_sat), the hash clamps (compute:126), the unclamped search cell
    (compute:155) finds no cell in range.  Hash, bins, neighbour lists and counters must be the oracle's bit for bit; the
    boundary pass then brings the finite ones back into the box."""
    se, so, eng, ora = _pair(lambda e: scenes.config1(e, n=4096))
    se.sph.use_step_frame = True
    se.start()
    so.start()
    state = ora.download_particles()
    big = [1e9, -1e9, 3e38, -3e38, np.inf, -np.inf, np.nan, 2147483648.0 * 0.25 - 5.0]
    k = 0
    for axis in range(3):
        for v in big:
            state[100 + k, axis] = v
            k += 1
    ora.upload_particles(state)
    eng.upload_particles(state)
    for e in (eng, ora):                           # grid build on the uploaded positions, kernel by kernel
        e.set_phase_table(se.sph._phase_rows())
        e.apply_phase_table()
        e.dispatch(abi.K_CLEAR_GRID, e.cells)
        e.dispatch(abi.K_BUILD_GRID, e.n)
    assert_grid_equal(eng.download_grid(), ora.download_grid(with_counts=True), so.sph.maxParticlesPerCell)
    assert_neighbors_equal(eng, ora, 512)
    assert eng.read_counter(abi.CTR_OUT_OF_GRID) == ora.read_counter(abi.CTR_OUT_OF_GRID) >= 21
    se.frame()
    so.frame()
    assert_grid_equal(eng.download_grid(), ora.download_grid(with_counts=True), so.sph.maxParticlesPerCell)
    got, want = eng.download_particles(), ora.download_particles()
    fin = np.isfinite(want).all(axis=1)
    assert np.array_equal(np.isfinite(got).all(axis=1), fin)
    assert_fields_close(got[fin], want[fin], 1e-5)
    eng.destroy()
