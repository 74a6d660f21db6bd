"""Kernel-granular lock-step: engine and oracle are driven by ONE host driver (tests.util.Pair), so mid-frame artefacts
can be compared.  -m gpu.

  * the hit mask: CS_DensityPressure's engine kernel leaves one bit per candidate slot and CS_ForceXSPH's engine kernel
    walks the set bits instead of testing the candidates again (compute:170-181 and 241-246 evaluated once).  The mask is
    the neighbour list the force kernel actually consumes; here it is decoded on the device (sph_download_hit_lists) and
    held bit-exactly, in visiting order, to the oracle's neighbour lists -- behind every CS_DensityPressure dispatch.
    Where a bin overflowed its capacity (rule O2) the engine must have recorded nothing (the capped scan path runs).
  * sub-steps: with subSteps > 1 the engine is re-synchronised to the oracle behind every sub-step
    (sph_upload_particles_in_place: SetData in the middle of a frame keeps the grid, as in the reference), so every
    sub-step is held to the plain 1e-5 of field scale, not only whole frames.
"""
import numpy as np
import pytest

from sph_amd import abi, scenes
from tests.test_random_scenes_gpu import _build, _draw
from tests.util import run_stepwise

pytestmark = pytest.mark.gpu
TOL = 1e-5          # of the field's scale, per sub-step


def test_hit_mask_scene2_two_phase():
    run_stepwise(lambda e: scenes.unity_scene(e, which=2, scale_count=0.02), 4, want_overflow=False)


def test_hit_mask_scene4_obstacle_two_sub_steps():
    def builder(e):
        sc = scenes.unity_scene(e, which=4, scale_count=0.02)
        sc.sph.subSteps = 2
        return sc
    st = run_stepwise(builder, 3, want_overflow=False)
    assert st["substeps"] == 6


def test_hit_mask_config3_small_with_both_movers():
    """Config 3 in small (rocking cube -> regrid every frame, orbiting sphere): the mask follows the moving grid."""
    run_stepwise(lambda e: scenes.config3(e, n=6000), 8, want_overflow=False)


def test_hit_mask_config1_dense():
    run_stepwise(lambda e: scenes.config1(e, n=16384), 2, want_overflow=False)


def _seeds(pred, k):
    return [s for s in range(24) if pred(_draw(s))][:k]


@pytest.mark.parametrize("seed", _seeds(lambda d: d["cap"] == 500, 2) + _seeds(lambda d: d["cap"] == 12, 1))
def test_hit_mask_random_scenes(seed):
    run_stepwise(lambda e: _build(e, _draw(seed)), 3, min_masked=0)


@pytest.mark.parametrize("seed", _seeds(lambda d: d["cap"] == 3, 24))
def test_hit_mask_absent_when_a_bin_overflows(seed):
    """cap = 3: bins overflow, the oracle's loops see the first three ids of a bin (rule O2); the engine must not have
    recorded a mask for such a frame (the capped 27-range scan runs), and where no bin overflowed the mask must match."""
    st = run_stepwise(lambda e: _build(e, _draw(seed)), 3, min_masked=0)
    assert st["overflow_masks"] + st["masks"] > 0


def test_some_cap3_case_really_overflows():
    from oracle.ora import Oracle
    hit = 0
    for seed in _seeds(lambda d: d["cap"] == 3, 24):
        ora = Oracle()
        sc = _build(ora, _draw(seed))
        sc.start()
        sc.frame()
        hit += ora.read_counter(abi.CTR_BIN_OVERFLOW) > 0
        ora.destroy()
    assert hit > 0


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 513])
def test_ragged_particle_counts_around_wave_and_workgroup_sizes(n):
    """One particle, a wave less / exactly / more than one, a workgroup less / exactly / more than one: the last
    workgroup is ragged (per-lane mask rows instead of the transposed wave copy, identity tile mapping), lanes past
    the end must stay out of ballots and rankings.  Kernel by kernel against the oracle, hit mask included, plus one
    sph_step_frame frame from the same state on a second engine (same bits as the dispatched frame)."""
    from sph_amd.host import SPH, Scene, SpawnBox, Time

    def builder(e):
        # ~6 particles per cell at any n: a small dense blob dropping onto the floor of the default AABB
        side = max((n / 400.0) ** (1.0 / 3.0), 0.12)
        boxes = [SpawnBox(n, (side, side, side), 20, 0.1, 0.2, position=(0.0, 0.6 * side + 0.05, 0.0), initialVelocity=(0.3, -1.0, 0.2))]
        f = dict(scenes.SCENE_COMMON)
        f.update(gridResolution=24)
        sph = SPH(e, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, spawnBoxes=boxes)
        return Scene(sph, Time(), [])

    st = run_stepwise(builder, 3, tol=TOL, max_nb=512, want_overflow=False)
    assert st["masks"] == 3
    a, b = abi.Engine(0), abi.Engine(0)
    sa, sb = builder(a), builder(b)
    sa.sph.use_step_frame, sb.sph.use_step_frame = True, False
    sa.start(); sb.start()
    for f in range(3):
        sa.frame(); sb.frame()
        assert np.array_equal(a.download_particles().view(np.uint32), b.download_particles().view(np.uint32)), f"frame {f}"
    a.destroy(); b.destroy()
