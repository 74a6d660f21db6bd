"""Host-side logic mirrored from Assets/SPH.cs / SpawnBox.cs / SphereMover.cs (CPU only)."""
import math

import numpy as np
import pytest

from sph_amd import abi, scenes
from sph_amd.host import (SPH, CircularMotion, SpawnBox, SphereCollider, Time, Transform, quat_angle,
                          quat_from_axis_angle, spawn_uniforms)


class RecordingEngine:
    """Stands in for ComputeShader/ComputeBuffer and records the call sequence."""
    grid_first = False

    def __init__(self):
        self.calls = []
        self.n = 0

    def __getattr__(self, name):
        def f(*a):
            self.calls.append((name,) + tuple(float(x) if isinstance(x, (float, np.floating)) else int(x)
                                               for x in a if isinstance(x, (int, float, np.integer, np.floating))))
            if name == "alloc_particles":
                self.n = a[0]
        return f


def test_update_dispatch_order_matches_sph_cs():
    e = RecordingEngine()
    sc = scenes.unity_scene(e, which=2, scale_count=0.001)
    sc.sph.subSteps = 2
    sc.start()
    e.calls.clear()
    sc.frame()
    disp = [c[1] for c in e.calls if c[0] == "dispatch"]
    sub = [abi.K_VV_HALF_STEP, abi.K_CLEAR, abi.K_DENSITY_PRESSURE, abi.K_FORCE_XSPH, abi.K_VV_FULL_STEP, abi.K_BOUND_OBS]
    assert disp == [abi.K_CLEAR_GRID, abi.K_BUILD_GRID] + sub + sub + [abi.K_VELOCITY_DAMPING]     # SPH.cs:201-222
    names = [c[0] for c in e.calls]
    assert names.index("reset_counter") < names.index("set_params") < names.index("set_phase_table") < names.index("dispatch")
    dts = [c for c in e.calls if c[0] == "set_delta_time"]
    assert len(dts) == 1 and dts[0][1] == pytest.approx(0.01)                                      # SPH.cs:207-208
    counts = [c[2] for c in e.calls if c[0] == "dispatch"]
    assert counts[0] == sc.sph.totalCells and all(c == sc.sph.particleCount for c in counts[1:])


def test_grid_sizing_from_boundary_cube():
    e = RecordingEngine()
    sc = scenes.unity_scene(e, which=1, scale_count=0.001)
    sc.start()
    s = sc.sph
    assert (s.gridResolutionX, s.gridResolutionY, s.gridResolutionZ, s.totalCells) == (80, 120, 80, 768000)
    assert float(s.cellSize) == 0.25 and s.gridMin.tolist() == [-10.0, 0.0, -10.0]
    assert ("resize_grid", 80, 120, 80, 500) in e.calls


def test_grid_sizing_aabb_fallback():
    e = RecordingEngine()
    sc = scenes.config1(e, n=64)
    sc.start()
    s = sc.sph
    assert (s.gridResolutionX, s.gridResolutionY, s.gridResolutionZ) == (40, 40, 40)
    assert s.gridMin.tolist() == [-5.0, 0.0, -5.0]
    p = s.build_params()
    assert list(p.boundaryHalfExtents) == [0.0, 0.0, 0.0]         # AABB mode flag (compute:349)
    assert list(p.boundsSize) == [10.0, 10.0, 10.0]


def test_regrid_only_when_cube_moves_beyond_threshold():
    e = RecordingEngine()
    sc = scenes.unity_scene(e, which=1, scale_count=0.001)
    sc.start()
    n0 = sum(1 for c in e.calls if c[0] == "resize_grid")
    sc.frame()
    assert sum(1 for c in e.calls if c[0] == "resize_grid") == n0          # nothing moved
    sc.sph.boundaryCube.position[0] += 0.0005                             # below 0.001 (SPH.cs:100)
    sc.frame()
    assert sum(1 for c in e.calls if c[0] == "resize_grid") == n0
    sc.sph.boundaryCube.rotation = quat_from_axis_angle((0, 0, 1), 5.0)
    sc.frame()
    assert sum(1 for c in e.calls if c[0] == "resize_grid") == n0 + 1      # SPH.cs:176-185
    assert sc.sph.gridResolutionX > 80                                     # rotated cube -> larger AABB


def test_quaternion_angle_dead_zone():
    a = quat_from_axis_angle((0, 0, 1), 0.0)
    assert quat_angle(a, quat_from_axis_angle((0, 0, 1), 0.05)) == 0.0     # Unity returns 0 below ~0.16 deg
    assert quat_angle(a, quat_from_axis_angle((0, 0, 1), 10.0)) == pytest.approx(10.0, abs=1e-3)


def test_trs_matrix_and_inverse():
    t = Transform(position=(1, 2, 3), rotation=quat_from_axis_angle((0, 1, 0), 90.0), localScale=(2, 3, 4))
    M = t.localToWorldMatrix
    assert M[:3, 3].tolist() == [1.0, 2.0, 3.0]
    assert t.TransformPoint((1, 0, 0)).tolist() == pytest.approx([1.0, 2.0, 3.0 - 2.0], abs=1e-6)
    assert np.allclose(M.astype(np.float64) @ t.worldToLocalMatrix.astype(np.float64), np.eye(4), atol=1e-6)


def test_circular_motion_orbit():
    time = Time()
    tr = Transform()
    m = CircularMotion(tr, time, center=(0, 0, 0), radius=5.0, speed=2.0)
    m.Update()
    assert tr.position.tolist() == [5.0, 0.0, 0.0]                         # frame 0 at t = 0 (rule O8)
    for _ in range(30):
        time.advance()
    m.Update()                                                             # t = 0.5 s, angle 1 rad
    assert tr.position.tolist() == pytest.approx([5 * math.cos(1.0), 0.0, 5 * math.sin(1.0)], abs=1e-5)


def test_obstacle_radius_uses_lossy_scale_x():
    e = RecordingEngine()
    s = SPH(e, obstacleCollider=SphereCollider(radius=0.5, transform=Transform(position=(1, 2, 3), localScale=(8, 1, 1))))
    s.UpdateObstacle()
    assert float(s._obstacleRadius) == 4.0 and s._obstaclePos.tolist() == [1.0, 2.0, 3.0]
    s.obstacleCollider = None
    s.UpdateObstacle()
    assert float(s._obstacleRadius) == 0.0                                 # SPH.cs:356-360


def test_spawn_rng_is_counter_based_and_in_box():
    u = spawn_uniforms(12345, 0, 1000)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
    assert np.array_equal(spawn_uniforms(12345, 500, 10), u[500:510])      # order independent
    assert not np.array_equal(spawn_uniforms(12346, 0, 10), u[:10])
    assert abs(float(u.mean()) - 0.5) < 0.02


def test_create_particles_layout_and_phase_ranges():
    class Capture(RecordingEngine):
        def upload_particles(self, a):
            self.aos = np.array(a)

        def set_phase_ids(self, ids):
            self.ids = np.array(ids)
    e = Capture()
    sc = scenes.unity_scene(e, which=2, scale_count=0.001)
    sc.start()
    P, ids = e.aos, e.ids
    assert P.shape == (225, 18) and ids.tolist() == [0] * 150 + [1] * 75   # contiguous per box (SPH.cs:253-279)
    assert np.all(P[:150, 11] == 20.0) and np.all(P[150:, 11] == 10.0)
    assert np.all(P[:150, 12] == np.float32(0.01)) and np.all(P[:, 13] == np.float32(0.2))
    assert np.all(P[:, 9] == P[:, 11]) and np.all(P[:, 10] == 0) and np.all(P[:, 3:9] == 0)
    c0 = np.array([0, 20, 5], np.float32)
    assert np.all(np.abs(P[:150, 0:3] - c0) <= np.array([2.5, 5, 2.5]) * 0.1 + 1e-6)
