"""Analytic known-answer tests that pin the CPU oracle.  The reference
ships no golden vectors, so closed-form answers are the first pin of the restatement."""
import math

import numpy as np
import pytest

from sph_amd import abi, scenes
from sph_amd.host import SPH, Scene, SpawnBox, Time, Transform

f32 = np.float32


@pytest.fixture()
def ora(oracle_built):
    from oracle.ora import Oracle
    o = Oracle()
    yield o
    o.destroy()


def one_box_scene(engine, positions, velocities=None, cube=None, rho0=20.0, mass=0.2, **over):
    n = len(positions)
    boxes = [SpawnBox(n, (0, 0, 0), rho0, 0.1, mass, position=(0, 5, 0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=40)
    f.update(over)
    sph = SPH(engine, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=cube, spawnBoxes=boxes)
    sc = Scene(sph, Time(), [])
    sc.start()
    P = engine.download_particles()
    P[:, 0:3] = np.asarray(positions, dtype=np.float32)
    if velocities is not None:
        P[:, 3:6] = np.asarray(velocities, dtype=np.float32)
    engine.upload_particles(P)
    return sc


def test_derived_constants(ora):
    one_box_scene(ora, [(0, 5, 0)])
    c = ora.constants()
    h = 0.3
    assert c["h2"] == pytest.approx(0.09, rel=1e-6)
    assert c["poly6"] == pytest.approx(315.0 / (64 * math.pi * h ** 9), rel=2e-6)            # ~7.9596e4
    assert c["spikyGradFactor"] == pytest.approx(-45.0 / (math.pi * h ** 6), rel=2e-6)       # ~-1.9649e4
    assert c["viscLapFactor"] == pytest.approx(45.0 / (math.pi * h ** 6), rel=2e-6)
    assert c["lapColFactor"] == pytest.approx(1890.0 / (64 * math.pi * h ** 9), rel=2e-6)    # ~4.7757e5
    assert c["B1"] * 20.0 == pytest.approx(500.0)                                            # B(rho0=20)


def test_cell_index_kat(ora):
    """pos (0,15,0), min (-10,0,-10), cell 0.25, 80x120x80 -> (40,60,40) -> 388,840."""
    cube = Transform(position=(0, 15, 0), localScale=(20, 30, 20))
    sc = one_box_scene(ora, [(0, 15, 0), (10.0, 15, 0), (-10.0, 0.0, -10.0)], cube=cube, gridResolution=80)
    assert (sc.sph.gridResolutionX, sc.sph.gridResolutionY, sc.sph.gridResolutionZ) == (80, 120, 80)
    assert float(sc.sph.cellSize) == 0.25
    ora.dispatch(abi.K_CLEAR_GRID)
    ora.dispatch(abi.K_BUILD_GRID)
    cop = ora.download_grid()[3]
    assert cop[0] == 388840
    assert cop[1] == 79 + 60 * 80 + 40 * 9600          # max face: coord 80 clamps to 79 (compute:126)
    assert cop[2] == 0
    assert ora.read_counter(abi.CTR_OUT_OF_GRID) == 1


def test_max_face_particle_searches_one_layer(ora):
    """Rule O7: a particle on the +x face has unclamped cell 80 and only sees x = 79."""
    cube = Transform(position=(0, 15, 0), localScale=(20, 30, 20))
    # j sits in cell x=78, 0.28 < h away from i but outside i's searched layer {79}
    one_box_scene(ora, [(10.0, 15.1, 0.1), (9.72, 15.1, 0.1), (9.8, 15.1, 0.1)], cube=cube, gridResolution=80)
    ora.dispatch(abi.K_CLEAR_GRID)
    ora.dispatch(abi.K_BUILD_GRID)
    cnt, lst = ora.download_neighbors(8)
    assert lst[0, :cnt[0]].tolist() == [2]               # cell 79 only
    assert sorted(lst[2, :cnt[2]].tolist()) == [0, 1]    # from cell 79 both sides are visible


def test_two_particle_density_and_pressure(ora):
    r = 0.2
    one_box_scene(ora, [(0, 5, 0), (r, 5, 0)])
    ora.dispatch(abi.K_CLEAR_GRID)
    ora.dispatch(abi.K_BUILD_GRID)
    ora.dispatch(abi.K_CLEAR)
    ora.dispatch(abi.K_DENSITY_PRESSURE)
    P = ora.download_particles()
    poly6 = 315.0 / (64 * math.pi * 0.3 ** 9)
    expect = max(0.2 * poly6 * (0.09 - r * r) ** 3, 20.0)
    assert P[0, 9] == pytest.approx(expect, rel=1e-5)
    assert P[1, 9] == pytest.approx(expect, rel=1e-5)
    assert P[0, 10] == pytest.approx(500.0 * ((expect / 20.0) ** 4 - 1.0), rel=1e-4, abs=1e-3)


def test_density_floor_gives_exactly_zero_pressure(ora):
    one_box_scene(ora, [(0, 5, 0), (0.29, 5, 0)])        # tiny kernel weight -> floored at rho0
    ora.dispatch(abi.K_CLEAR_GRID)
    ora.dispatch(abi.K_BUILD_GRID)
    ora.dispatch(abi.K_DENSITY_PRESSURE)
    P = ora.download_particles()
    assert P[0, 9] == 20.0 and P[0, 10] == 0.0


def test_single_particle_free_fall_frame(ora):
    """No neighbours: rho = rho0, p = 0, a = (0,-9.81,0); v_y = -0.5*0.02*9.81 then x0.98."""
    sc = one_box_scene(ora, [(0, 5, 0)])
    sc.frame()
    P = ora.download_particles()[0]
    assert P[9] == 20.0 and P[10] == 0.0
    assert P[6:9].tolist() == pytest.approx([0.0, -9.81, 0.0], rel=1e-6)
    assert np.array_equal(P[0:3], np.array([0, 5, 0], dtype=np.float32))        # half step moved nothing
    vy = f32(f32(0.5) * f32(0.02)) * f32(-9.81)
    assert P[4] == pytest.approx(float(f32(vy * f32(0.98))), rel=1e-6)           # -0.096138
    assert P[4] == pytest.approx(-0.096138, rel=1e-5)


def test_velocity_damping_threshold(ora):
    sc = one_box_scene(ora, [(0, 5, 0), (3, 5, 0)], velocities=[(0.1, 0, 0), (0.3, 0, 0)])
    ora.set_params(sc.sph.build_params())
    ora.dispatch(abi.K_VELOCITY_DAMPING)
    P = ora.download_particles()
    assert P[0, 3] == f32(f32(0.1) * f32(0.98))
    assert P[1, 3] == f32(f32(0.3) * f32(0.999))


def test_obb_corner_normal_uses_matrix_not_inverse_transpose(ora):
    """Particle beyond two faces of the (20,30,20) cube: local normal (1,1,0)/sqrt2, world normal
    proportional to (20,30,0) (compute:408 multiplies by M, a reference property we keep)."""
    cube = Transform(position=(0, 15, 0), localScale=(20, 30, 20))
    sc = one_box_scene(ora, [(10.5, 30.5, 0.0)], velocities=[(1.0, 2.0, 3.0)], cube=cube, gridResolution=80)
    ora.set_params(sc.sph.build_params())
    ora.reset_counter(abi.CTR_COLLISIONS)
    ora.dispatch(abi.K_BOUND_OBS)
    P = ora.download_particles()[0]
    assert P[0:3].tolist() == pytest.approx([10.0, 30.0, 0.0], abs=1e-5)
    n = np.array([20.0, 30.0, 0.0]) / math.hypot(20.0, 30.0)
    v = np.array([1.0, 2.0, 3.0])
    assert P[3:6].tolist() == pytest.approx((v - 2 * v.dot(n) * n).tolist(), rel=1e-5)
    assert ora.read_counter(abi.CTR_COLLISIONS) == 1


def test_aabb_flips_only_outward_velocity(ora):
    sc = one_box_scene(ora, [(-5.5, 5, 0), (5.5, 5, 0)], velocities=[(1.0, 0, 0), (1.0, 0, 0)])
    ora.set_params(sc.sph.build_params())
    ora.dispatch(abi.K_BOUND_OBS)
    P = ora.download_particles()
    assert P[0, 0] == -5.0 and P[0, 3] == 1.0                     # moving inward: untouched
    assert P[1, 0] == 5.0 and P[1, 3] == f32(f32(-1.0) * f32(0.9))  # moving outward: flipped, x0.9


def test_sphere_obstacle_push_out_and_reflect(ora):
    from sph_amd.host import SphereCollider
    sc = one_box_scene(ora, [(1.0, 5.0, 0.0), (0.0, 5.0, 0.0)], velocities=[(-1.0, 0.5, 0), (0.0, -2.0, 0.0)])
    sc.sph.obstacleCollider = SphereCollider(radius=0.5, transform=Transform(position=(0, 5, 0), localScale=(4, 4, 4)))
    sc.sph.particleCollisionDamping = 0.5
    sc.sph.UpdateObstacle()
    ora.set_params(sc.sph.build_params())
    ora.dispatch(abi.K_BOUND_OBS)
    P = ora.download_particles()
    assert P[0, 0:3].tolist() == pytest.approx([2.0, 5.0, 0.0], abs=1e-6)       # pushed to the surface
    assert P[0, 3:6].tolist() == pytest.approx([0.5, 0.25, 0.0], abs=1e-6)      # reflect(v,(1,0,0)) * 0.5
    assert P[1, 1] == pytest.approx(5.0 + 2.0 + 0.01)                           # centre case (compute:478-482)
    assert P[1, 4] == pytest.approx(1.0)


def test_bin_cap_keeps_first_ids(ora):
    """Rule O2: GridCounts counts everyone, the bin keeps the first cap ids by id."""
    pts = [(0.01 * k, 5.0, 0.0) for k in range(12)]      # all in one cell
    one_box_scene(ora, pts, maxParticlesPerCell=8)
    ora.dispatch(abi.K_CLEAR_GRID)
    ora.dispatch(abi.K_BUILD_GRID)
    cs, ce, ids, cop, gc = ora.download_grid(with_counts=True)
    c = int(cop[0])
    assert gc[c] == 12 and ce[c] - cs[c] == 8
    assert ids[cs[c]:ce[c]].tolist() == list(range(8))
    assert ora.read_counter(abi.CTR_BIN_OVERFLOW) == 1
    cnt, lst = ora.download_neighbors(16)
    assert lst[11, :cnt[11]].tolist() == list(range(8))  # particle 11 is not in the bin but still searches it


def test_xsph_jacobi_vs_inplace_differ_but_little(ora):
    """Rule O3 sensitivity: serial in-place XSPH differs from the Jacobi parity target."""
    from oracle.ora import Oracle
    a, b = Oracle(), Oracle()
    b.set_xsph_inplace(1)
    sa, sb = scenes.config1(a, n=512), scenes.config1(b, n=512)
    for s in (sa, sb):
        s.sph.spawnBoxes[0].spawnSize[:] = (1.0, 1.0, 1.0)
        s.sph.spawnBoxes[0].initialVelocity[:] = (0.5, 0.0, 0.0)
    sa.run(3); sb.run(3)
    va, vb = a.download_particles()[:, 3:6], b.download_particles()[:, 3:6]
    assert not np.array_equal(va, vb)
    assert np.abs(va - vb).max() < 0.2 * np.abs(va).max()


def test_oracle_is_clean_under_asan_and_ubsan():
    """SURVEY section 5 (race detection / sanitizers): the oracle and a driver that walks its memory-indexing branches
    (bin overflow, particles outside the grid on every side, NaN, OBB / AABB / sphere, grid resizes) built with
    -fsanitize=address,undefined; any report fails `make -C oracle asan`."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:      # is there a sanitizer runtime to link against at all?
        src = os.path.join(tmp, "t.cpp")
        open(src, "w").write("int main() { return 0; }\n")
        probe = subprocess.run(["g++", "-fsanitize=address,undefined", src, "-o", os.path.join(tmp, "t")], capture_output=True, text=True)
        if probe.returncode != 0:
            pytest.skip("libasan / libubsan are not installed here: " + probe.stderr.strip().splitlines()[-1][:200])
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok") >= 2 and "FAILED" not in r.stdout
