"""The two routes through CS_ForceXSPH in libsph_hip -- walking the hit mask left by
CS_DensityPressure (k_force_xsph_hits) and re-testing every candidate (k_force_xsph_rows / the
per-wave fallback) -- must give the same bits, and both must match the oracle.  -m gpu."""
import os

import numpy as np
import pytest

from sph_amd import abi, scenes
from sph_amd.host import SPH, Scene, SpawnBox, Time
from tests.util import assert_fields_close, assert_neighbors_equal

pytestmark = pytest.mark.gpu


def engine_without_hit_mask():
    os.environ["SPH_HIP_HITMASK"] = "0"          # read once, in sph_create
    try:
        return abi.Engine(0)
    finally:
        del os.environ["SPH_HIP_HITMASK"]


def crowded(e, n=3000, cap=500):
    """~90 particles per cell, bins not capped: every particle has > 512 candidates, more than the
    hit mask holds, so k_force_xsph_hits takes its per-wave fallback."""
    boxes = [SpawnBox(n, (0.8, 0.8, 0.8), 20, 0.1, 0.2, position=(0.0, 5.0, 0.0))]
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=40, maxParticlesPerCell=cap)
    sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, spawnBoxes=boxes)
    return Scene(sph, Time(), [])


@pytest.mark.parametrize("step_frame", [False, True])
def test_hit_path_and_scan_path_agree_bitwise(step_frame):
    a, b = abi.Engine(0), engine_without_hit_mask()
    sa, sb = scenes.unity_scene(a, which=4, scale_count=0.05), scenes.unity_scene(b, which=4, scale_count=0.05)
    for s in (sa, sb):
        s.sph.use_step_frame = step_frame
        s.start()
    for f in range(6):
        sa.frame(); sb.frame()
        pa, pb = a.download_particles(), b.download_particles()
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), f"frame {f}"
    a.destroy(); b.destroy()


@pytest.mark.parametrize("n", [3000, 700])
def test_crowded_particles_fall_back_and_match_the_oracle(n):
    """n = 3000: more candidates than any mask holds (per-wave fallback to the scan path);
    n = 700: ~21 particles per cell, rows of ~64 candidates = two mask words each, ~18 words per particle:
    the second (overflow) mask array is in use."""
    from oracle.ora import Oracle
    eng, ref, ora = abi.Engine(0), engine_without_hit_mask(), Oracle()
    se, sr, so = crowded(eng, n), crowded(ref, n), crowded(ora, n)
    for s in (se, sr, so):
        s.sph.use_step_frame = s is not so
        s.start()
    for f in range(2):
        state = ora.download_particles()
        eng.upload_particles(state); ref.upload_particles(state)
        if f == 0:   # particles are at rest at spawn: the frame's neighbour search sees this crowd
            x = state[:, 0:3].astype(np.float64)
            d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
            if n == 3000:
                assert ((d2 < 0.09).sum(1) - 1).max() > 512  # hits alone; candidates are ~4x that, far beyond 24 mask words
        se.frame(); sr.frame(); so.frame()
        assert eng.read_counter(abi.CTR_BIN_OVERFLOW) == 0
        assert_neighbors_equal(eng, ora, 4096, f"frame {f}: ")
        pe = eng.download_particles()
        assert np.array_equal(pe.view(np.uint32), ref.download_particles().view(np.uint32)), f"frame {f}"
        assert_fields_close(pe, ora.download_particles(), 1e-5, f"frame {f}: ")
    eng.destroy(); ref.destroy(); ora.destroy()


def test_force_dispatch_without_a_fresh_mask():
    """Kernel-granular callers may dispatch CS_ForceXSPH after something moved the particles
    since the last CS_DensityPressure: the library must not trust the stale mask."""
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle()
    se, so = scenes.config1(eng, n=4096), scenes.config1(ora, n=4096)
    se.start(); so.start()
    se.frame(); so.frame()
    eng.upload_particles(ora.download_particles())
    for e, s in ((eng, se), (ora, so)):
        n, cells = e.n, s.sph.gridResolutionX * s.sph.gridResolutionY * s.sph.gridResolutionZ
        e.dispatch(abi.K_CLEAR_GRID, cells)
        e.dispatch(abi.K_BUILD_GRID, n)
        e.dispatch(abi.K_DENSITY_PRESSURE, n)
        e.dispatch(abi.K_VV_HALF_STEP, n)           # positions move: mask is stale now
        e.dispatch(abi.K_FORCE_XSPH, n)
    assert_fields_close(eng.download_particles(), ora.download_particles(), 1e-5, "stale mask: ")
    eng.destroy(); ora.destroy()


def test_live_phase_edit_lockstep():
    """SPH.cs:284-298 re-applies the spawn boxes' restDensity / viscosity / mass every frame, so editing
    a box mid-run changes the particles of that phase.  The engine's packed neighbour records look
    mass and viscosity up by phase: they must follow the edit exactly like the oracle's per-particle
    fields do."""
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle()
    se, so = scenes.unity_scene(eng, which=4, scale_count=0.03), scenes.unity_scene(ora, which=4, scale_count=0.03)
    se.sph.use_step_frame = True
    se.start(); so.start()
    for f in range(6):
        if f == 2:
            for sc in (se, so):
                sc.sph.spawnBoxes[1].particleMass = 0.25
                sc.sph.spawnBoxes[1].viscosity = 0.07
        if f == 4:
            for sc in (se, so):
                sc.sph.spawnBoxes[0].viscosity = 0.2
        eng.upload_particles(ora.download_particles())
        se.frame(); so.frame()
        pe, po = eng.download_particles(), ora.download_particles()
        assert np.array_equal(pe[:, 11:14], po[:, 11:14]), f"frame {f}: static fields"
        assert_fields_close(pe, po, 1e-5, f"frame {f}: ")
    eng.destroy(); ora.destroy()


def engine_without_pos3():
    os.environ["SPH_HIP_UNIFORM_MASS"] = "0"    # read once, in sph_create
    try:
        return abi.Engine(0)
    finally:
        del os.environ["SPH_HIP_UNIFORM_MASS"]


@pytest.mark.parametrize("step_frame", [True, False])
def test_density_12_byte_candidates_equal_16_byte_bitwise(step_frame):
    """Same mass in every phase: the density kernel reads candidate positions from the 12-byte array the half step
    left (k_density_pressure_rows<true>).  Same bits as the 16-byte posM path, frame after frame."""
    a, b = abi.Engine(0), engine_without_pos3()
    sa, sb = scenes.unity_scene(a, which=4, scale_count=0.05), scenes.unity_scene(b, which=4, scale_count=0.05)
    sa.sph.use_step_frame = sb.sph.use_step_frame = step_frame
    sa.sph.subSteps = sb.sph.subSteps = 2          # the second sub-step's positions come from k_half_step
    sa.start(); sb.start()
    for f in range(4):
        sa.frame(); sb.frame()
        x, y = a.download_particles(), b.download_particles()
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"frame {f}"
    a.destroy(); b.destroy()


def test_phases_of_different_mass_take_the_16_byte_path_and_match_the_oracle():
    """Different particle masses per phase: no uniform-mass shortcut; density still bit-identical to the oracle."""
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle()
    se, so = scenes.unity_scene(eng, which=2, scale_count=0.02), scenes.unity_scene(ora, which=2, scale_count=0.02)
    for sc in (se, so):
        sc.sph.spawnBoxes[1].particleMass = 0.15
    se.sph.use_step_frame = True
    se.start(); so.start()
    for f in range(3):
        eng.upload_particles(ora.download_particles())
        se.frame(); so.frame()
        x, y = eng.download_particles(), ora.download_particles()
        assert np.array_equal(x[:, 9].view(np.uint32), y[:, 9].view(np.uint32)), f"frame {f}: density"
        assert_fields_close(x, y, 1e-5, f"frame {f}: ")
    eng.destroy(); ora.destroy()


def test_uniform_mass_density_is_bitwise_the_oracles():
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle()
    se, so = scenes.unity_scene(eng, which=2, scale_count=0.02), scenes.unity_scene(ora, which=2, scale_count=0.02)
    se.sph.use_step_frame = True
    se.start(); so.start()
    for f in range(3):
        eng.upload_particles(ora.download_particles())
        se.frame(); so.frame()
        x, y = eng.download_particles(), ora.download_particles()
        assert np.array_equal(x[:, 9].view(np.uint32), y[:, 9].view(np.uint32)), f"frame {f}: density"
    eng.destroy(); ora.destroy()


def engine_with_env(**env):
    """An engine created under the given SPH_HIP_* switches (all of them are read once, in sph_create)."""
    for k, v in env.items():
        os.environ[k] = v
    try:
        return abi.Engine(0)
    finally:
        for k in env:
            del os.environ[k]


@pytest.mark.parametrize("which,sub", [(4, 1), (4, 2), (2, 1)])
def test_tail_inside_the_force_kernel_equals_the_separate_pass_bitwise(which, sub):
    """sph_step_frame on one GPU applies CS_VvFullStep + CS_BoundObs (+ CS_VelocityDamping) inside the force kernel
    (k_force_xsph_hits<MODE, TAIL=true>, new positions into the other position array).  Same bits as the separate
    tail pass (SPH_HIP_FUSE_TAIL=0) and as kernel-by-kernel dispatch, collision counter included, with obstacles
    and a moving boundary in the scene."""
    a, b, c = abi.Engine(0), engine_with_env(SPH_HIP_FUSE_TAIL="0"), abi.Engine(0)
    ss = [scenes.unity_scene(e, which=which, scale_count=0.05) for e in (a, b, c)]
    for s, sf in zip(ss, (True, True, False)):
        s.sph.use_step_frame = sf
        s.sph.subSteps = sub
        s.start()
    for f in range(6):
        for s in ss:
            s.frame()
        pa, pb, pc = a.download_particles(), b.download_particles(), c.download_particles()
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), f"frame {f}: fused vs separate tail"
        assert np.array_equal(pa.view(np.uint32), pc.view(np.uint32)), f"frame {f}: fused vs dispatch"
        assert a.read_counter(abi.CTR_COLLISIONS) == b.read_counter(abi.CTR_COLLISIONS) == c.read_counter(abi.CTR_COLLISIONS)
    a.destroy(); b.destroy(); c.destroy()


@pytest.mark.parametrize("n", [20000, 4096 + 77])
def test_lds_staged_density_equals_the_rows_kernel_bitwise(n, capfd):
    """SPH_HIP_LDS_DENSITY=1 (k_density_pressure_lds: workgroup windows of pos3 staged in LDS) is an opt-in experiment
    (profiles/r3_lds_density); it has to leave the same density, pressure, hit mask and neighbour records as the
    default kernel: same bits frame after frame, and the decoded hit lists equal the oracle's neighbour lists."""
    from oracle.ora import Oracle
    from tests.util import assert_hit_lists_equal
    a, b, ora = abi.Engine(0), engine_with_env(SPH_HIP_LDS_DENSITY="1", SPH_HIP_LDS_STATS="1"), Oracle()
    sa, sb, so = scenes.config1(a, n=n), scenes.config1(b, n=n), scenes.config1(ora, n=n)
    for s in (sa, sb):
        s.sph.use_step_frame = True
    for s in (sa, sb, so):
        s.start()
    for f in range(5):
        state = ora.download_particles()
        a.upload_particles(state); b.upload_particles(state)
        sa.frame(); sb.frame(); so.frame()
        pa, pb = a.download_particles(), b.download_particles()
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), f"frame {f}"
        assert np.array_equal(pb[:, 9].view(np.uint32), ora.download_particles()[:, 9].view(np.uint32)), f"frame {f}: density"
    # the mask itself: grid, half step (it leaves the 12-byte positions both kernels read) and density dispatched on
    # all three from the same state, then decoded
    for e, s in ((a, sa), (b, sb), (ora, so)):
        e.upload_particles(state)
        cells = s.sph.gridResolutionX * s.sph.gridResolutionY * s.sph.gridResolutionZ
        e.dispatch(abi.K_CLEAR_GRID, cells); e.dispatch(abi.K_BUILD_GRID, e.n)
        e.dispatch(abi.K_VV_HALF_STEP, e.n); e.dispatch(abi.K_DENSITY_PRESSURE, e.n)
    sa_, ca, la = a.download_hit_lists(256)
    sb_, cb, lb = b.download_hit_lists(256)
    assert sa_ == sb_ == 1
    assert np.array_equal(ca, cb) and np.array_equal(la, lb)
    assert_hit_lists_equal(b, ora, 256, "lds kernel: ")
    capfd.readouterr()
    a.destroy(); b.destroy(); ora.destroy()
    import re
    m = re.search(r"k_density_pressure_lds: lane-steps from LDS (\d+), from memory (\d+)", capfd.readouterr().err)
    assert m, "the LDS kernel's statistics line is missing: was it launched at all?"
    from_lds, from_memory = int(m.group(1)), int(m.group(2))
    # both routes of the kernel ran: most lane-steps from the staged windows, the uncovered rest from memory
    assert from_lds > 6 * 9 * n * 0.3 and from_memory < 0.5 * from_lds, (from_lds, from_memory)
