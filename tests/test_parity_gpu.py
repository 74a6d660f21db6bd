"""Lock-step parity of libsph_hip (through the C ABI) against the CPU oracle.  -m gpu."""
import numpy as np
import pytest

from sph_amd import abi, scenes
from tests.util import assert_fields_close, assert_grid_equal, assert_neighbors_equal

pytestmark = pytest.mark.gpu


def make_pair(builder, **kw):
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle()
    return builder(eng, **kw), builder(ora, **kw), eng, ora


def lockstep(builder, frames, tol=1e-5, check_nb=True, step_frame=False, **kw):
    """Run `frames` frames; before each one the engine is re-synchronised to the oracle's
    state (lock-step regime): same input state -> one frame -> compare."""
    se, so, eng, ora = make_pair(builder, **kw)
    se.sph.use_step_frame = step_frame
    se.start(); so.start()
    for f in range(frames):
        eng.upload_particles(ora.download_particles())
        se.frame(); so.frame()
        cap = so.sph.maxParticlesPerCell
        assert_grid_equal(eng.download_grid(), ora.download_grid(with_counts=True), cap, f"frame {f}: ")
        if check_nb:
            assert_neighbors_equal(eng, ora, 512, f"frame {f}: ")
        assert_fields_close(eng.download_particles(), ora.download_particles(), tol, f"frame {f}: ")
        for ctr in range(3):
            assert eng.read_counter(ctr) == ora.read_counter(ctr), f"frame {f}: counter {ctr}"
    return eng, ora


def test_config1_aabb_lockstep():
    lockstep(scenes.config1, 5, n=4096)


def test_config1_aabb_dense_lockstep():
    lockstep(scenes.config1, 3, n=16384)


def test_scene2_obb_two_phase_lockstep():
    lockstep(scenes.unity_scene, 4, which=2, scale_count=0.02)


def test_scene4_obstacle_lockstep():
    lockstep(scenes.unity_scene, 4, which=4, scale_count=0.02)


def test_step_frame_equals_dispatch_sequence():
    lockstep(scenes.unity_scene, 3, which=2, scale_count=0.02, step_frame=True)


@pytest.mark.parametrize("step_frame", [False, True])
def test_two_sub_steps_lockstep(step_frame):
    """subSteps = 2 (SPH.cs:207-219): the grid is built once, the second sub-step searches it with
    positions two half-steps old; sph_step_frame folds only the FIRST half step into the reorder."""
    def builder(e, **kw):
        sc = scenes.unity_scene(e, which=4, scale_count=0.02)
        sc.sph.subSteps = 2
        return sc
    lockstep(builder, 3, step_frame=step_frame)
