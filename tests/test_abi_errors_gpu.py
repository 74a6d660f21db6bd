"""Error behaviour and edge cases at the C ABI on the GPU: every misuse returns a negative SPH_E_* code with a message
(never throws, never aborts, never launches on bad shapes), and the smallest inputs behave like the analytic
known-answer tests of SURVEY 8(c).  -m gpu."""
import ctypes as C

import numpy as np
import pytest

from sph_amd import abi, scenes

pytestmark = pytest.mark.gpu

E_INVALID, E_STATE, E_NO_DEVICE = -1, -5, -4


def _code(fn, *a):
    with pytest.raises(abi.SphError) as ei:
        fn(*a)
    assert str(ei.value).split(": ", 1)[1], "error without a message"
    return ei.value.code


def test_create_on_a_missing_device():
    h = C.c_void_p()
    assert abi.load_library().sph_create(C.byref(h), 4096) == E_NO_DEVICE and not h.value
    assert abi.load_library().sph_create(C.byref(h), -1) == E_NO_DEVICE


def test_shapes_are_checked_before_anything_is_launched():
    e = abi.Engine(0)
    assert _code(e.alloc_particles, 0) == E_INVALID
    assert _code(e.alloc_particles, -5) == E_INVALID
    assert _code(e.alloc_particles, (1 << 28)) == E_INVALID              # beyond the 32-bit byte offsets of the kernels
    assert _code(e.resize_grid, 0, 4, 4, 100) == E_INVALID
    assert _code(e.resize_grid, 2048, 2048, 2048, 100) == E_INVALID       # 2^33 cells: refused, nothing allocated
    assert _code(e.step_frame, 1, 0.02) == E_STATE                        # nothing allocated yet
    e.alloc_particles(100)
    assert _code(e.upload_particles, np.zeros((99, 18), np.float32)) == E_INVALID
    assert e.lib.sph_upload_particles(e.h, None, 100) == E_INVALID
    assert e.lib.sph_download_particles(e.h, None, 100) == E_INVALID
    assert _code(e.dispatch, abi.K_BUILD_GRID, 100) == E_STATE            # no params, no grid
    e.resize_grid(8, 8, 8, 100)
    assert _code(e.step_frame, 1, 0.02) == E_STATE                        # sph_set_params not called
    p = abi.SphParams()
    p.struct_size = 12
    assert e.lib.sph_set_params(e.h, C.byref(p)) == E_INVALID             # a header of another version
    assert _code(e.set_phase_ids, np.zeros(100, np.uint8)) == E_STATE     # no table yet
    e.set_phase_table([(20.0, 0.1, 0.2, (1, 1, 1, 1))])
    assert _code(e.set_phase_ids, np.zeros(99, np.uint8)) == E_INVALID
    assert _code(e.set_phase_ids, np.ones(100, np.uint8)) == E_INVALID    # id 1 with a one-row table
    assert _code(e.read_counter, 99) == E_INVALID
    assert _code(e.device_pointer, "nope") == E_INVALID
    e.destroy()


def test_call_order_is_checked():
    e = abi.Engine(0)
    sc = scenes.config1(e, n=512)
    sc.start()
    n, cells = e.n, e.cells
    assert _code(e.dispatch, abi.K_DENSITY_PRESSURE, n) == E_STATE        # the reference would search stale bins
    assert _code(e.dispatch, abi.K_FORCE_XSPH, n) == E_STATE
    assert _code(e.dispatch, abi.K_BUILD_GRID, n - 1) == E_INVALID        # SPH.cs always dispatches the full buffer
    assert _code(e.dispatch, abi.K_CLEAR_GRID, cells + 1) == E_INVALID
    assert _code(e.dispatch, 9, n) == E_INVALID
    assert _code(e.step_frame, 0, 0.02) == E_INVALID
    assert _code(e.get_timings) == E_STATE                                # no timed frame yet
    e.dispatch(abi.K_CLEAR_GRID, cells)
    e.dispatch(abi.K_BUILD_GRID, n)
    e.dispatch(abi.K_DENSITY_PRESSURE, n)                                 # now legal
    e.upload_particles(e.download_particles())                            # an upload voids the grid again
    assert _code(e.dispatch, abi.K_DENSITY_PRESSURE, n) == E_STATE
    # the context is still usable after all those errors
    sc.sph.use_step_frame = True
    sc.frame()
    assert np.isfinite(e.download_particles()).all()
    e.destroy()


def _one_frame(rows, **fields):
    """rows: (k, 3) positions of k resting particles of one phase in the AABB-fallback box of config 1."""
    from sph_amd.host import SPH, Scene, SpawnBox, Time
    e = abi.Engine(0)
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=40)
    f.update(fields)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, spawnBoxes=[SpawnBox(len(rows), (1, 1, 1), 20, 0.1, 0.2)])
    sc = Scene(sph, Time(), [])
    sc.start()
    aos = e.download_particles()
    aos[:, 0:3] = np.asarray(rows, dtype=np.float32)
    e.upload_particles(aos)
    sc.sph.use_step_frame = True
    sc.frame()
    out = e.download_particles()
    ctr = [e.read_counter(k) for k in range(3)]
    e.destroy()
    return out, ctr


def test_single_particle_free_fall_known_answer():
    """no neighbours -> density = rest density, pressure 0; a = (0, g, 0); v_y = 0.5 dt g, then damped
    by 0.98 (|v| < 0.2): -0.096138."""
    out, ctr = _one_frame([(0.0, 5.0, 0.0)])
    p = out[0]
    assert p[9] == np.float32(20.0) and p[10] == np.float32(0.0)
    assert np.allclose(p[6:9], (0.0, -9.81, 0.0), rtol=0, atol=1e-6)
    want_vy = np.float32(np.float32(0.5) * np.float32(0.02) * np.float32(-9.81)) * np.float32(0.98)
    assert abs(p[4] - want_vy) <= 1e-7 and abs(p[4] + 0.096138) < 1e-6
    assert tuple(p[0:3]) == (0.0, 5.0, 0.0)                               # the half step moves nothing from rest
    assert ctr == [0, 0, 0]


def test_two_particle_density_known_answer():
    """rho = max(m poly6 (h^2 - r^2)^3, rho0) for a pair at distance r < h (compute:180-186)."""
    r = 0.1
    out, _ = _one_frame([(0.0, 5.0, 0.0), (r, 5.0, 0.0)])
    h = np.float32(0.3)
    poly6 = np.float32(315.0) / (np.float32(64.0) * np.float32(3.14159265359) * np.float32(np.power(h, np.float32(9.0))))
    d = np.float32(h * h) - np.float32(np.float32(r) * np.float32(r))
    want = max(np.float32(np.float32(0.2) * poly6 * d * d * d), np.float32(20.0))
    assert np.allclose(out[:, 9], want, rtol=2e-6)
    assert out[0, 9] == out[1, 9]


def test_particle_outside_the_grid_is_clamped_and_counted():
    """compute:125-127 clamps the cell of a particle outside the grid; the engine also counts it."""
    out, ctr = _one_frame([(0.0, 5.0, 0.0), (100.0, 5.0, 0.0)])
    assert ctr[abi.CTR_OUT_OF_GRID] == 1 and np.isfinite(out).all()
    # CS_BoundObs (AABB branch) has put it back on the +x face
    assert out[1, 0] == np.float32(5.0)


def test_bin_overflow_is_a_counter_not_an_error():
    rng = np.random.default_rng(3)
    rows = np.array([0.0, 5.0, 0.0]) + rng.uniform(0.01, 0.24, size=(40, 3))          # 40 particles in ONE cell
    out, ctr = _one_frame(rows, maxParticlesPerCell=16)
    assert ctr[abi.CTR_BIN_OVERFLOW] == 1 and np.isfinite(out).all()
