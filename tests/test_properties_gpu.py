"""Size-independent properties of the path at the size the metric is quoted on (BASELINE config 3, 1,048,576 particles),
checked without the oracle -- it takes seconds per frame there -- so that they can run over more frames than the
lock-step tests: the grid is a sorted partition (rule O1), building it is idempotent, and the whole frame is
deterministic run to run although the counting sort hands out bucket slots with atomics.  -m gpu."""
import numpy as np
import pytest

from sph_amd import abi, scenes

pytestmark = pytest.mark.gpu
N = 1048576


def _grid_is_a_sorted_partition(e, sph, what):
    """cell ranges tile [0, n) in cell order, every id occurs once, ids ascend inside every bin (compute:117-136 with
    rule O1), and every particle sits in the bin its position hashes to (compute:120-127, clamped)."""
    n = e.n
    cs, ce, ids, cop = e.download_grid()
    cells = sph.gridResolutionX * sph.gridResolutionY * sph.gridResolutionZ
    assert cs.shape[0] == cells and ce.shape[0] == cells
    assert cs[0] == 0 and ce[-1] == n and np.array_equal(cs[1:], ce[:-1]), what + "ranges do not tile [0, n)"
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=ids.dtype)), what + "sorted ids are not a permutation"
    # ascending ids inside every bin: a descent may only happen where a new bin starts
    starts = np.zeros(n + 1, dtype=bool)
    starts[cs[cs < n]] = True
    descents = np.flatnonzero(ids[1:] < ids[:-1]) + 1
    assert starts[descents].all(), what + "ids do not ascend inside a bin"
    # the bin of every particle is the bin of its hash
    cell_of_slot = np.repeat(np.arange(cells, dtype=np.int64), (ce - cs).astype(np.int64))
    assert np.array_equal(cell_of_slot, cop[ids].astype(np.int64)), what + "a particle sits in a bin that is not its hash"
    return cs, ce, ids, cop


def test_grid_build_is_a_sorted_partition_and_idempotent_at_full_size():
    e = abi.Engine(0)
    sc = scenes.config3(e, n=N)
    sc.sph.use_step_frame = True
    sc.start()
    for _ in range(6):
        sc.frame()
    sph = sc.sph
    cells = sph.gridResolutionX * sph.gridResolutionY * sph.gridResolutionZ
    before = e.download_particles()
    e.dispatch(abi.K_CLEAR_GRID, cells); e.dispatch(abi.K_BUILD_GRID, e.n)
    g1 = _grid_is_a_sorted_partition(e, sph, "first build: ")
    mid = e.download_particles()
    assert np.array_equal(before.view(np.uint32), mid.view(np.uint32)), "CS_BuildGrid changed a particle (id order export)"
    e.dispatch(abi.K_CLEAR_GRID, cells); e.dispatch(abi.K_BUILD_GRID, e.n)
    g2 = _grid_is_a_sorted_partition(e, sph, "second build: ")
    for a, b, name in zip(g1, g2, ("cell_start", "cell_end", "sorted_ids", "cell_of_particle")):
        assert np.array_equal(a, b), f"building the grid twice from the same positions changed {name}"
    assert np.array_equal(mid.view(np.uint32), e.download_particles().view(np.uint32))
    # the hash against the host's arithmetic for every particle (compute:120-127: floor((p - min) / cell), clamped)
    p = mid[:, 0:3]
    prm = sph.build_params()
    c = np.floor((p - np.asarray(prm.minBound[:3], dtype=np.float32)) / np.float32(prm.cellSize)).astype(np.int64)
    res = np.array([sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ], dtype=np.int64)
    c = np.clip(c, 0, res - 1)
    want = c[:, 0] + c[:, 1] * res[0] + c[:, 2] * res[0] * res[1]
    assert np.array_equal(want, g2[3].astype(np.int64)), "cell hash differs from floor((p - min) / cell) clamped"
    e.destroy()


@pytest.mark.parametrize("step_frame", [True, False])
def test_frames_are_deterministic_run_to_run_at_full_size(step_frame):
    """Two engines, the same scene, 12 frames (regrid every frame under the rocking cube): the counting sort's atomics
    arrive in another order every run, the (cell, id) ranking removes that, and every neighbour sum runs in visiting
    order -- so the two states are the same bits, collision and out-of-grid counters included."""
    states, counters = [], []
    for _ in range(2):
        e = abi.Engine(0)
        sc = scenes.config3(e, n=N)
        sc.sph.use_step_frame = step_frame
        sc.start()
        for _ in range(12 if step_frame else 5):
            sc.frame()
        states.append(e.download_particles())
        counters.append([e.read_counter(k) for k in (abi.CTR_COLLISIONS, abi.CTR_BIN_OVERFLOW, abi.CTR_OUT_OF_GRID, abi.CTR_NAN)])
        e.destroy()
    assert np.array_equal(states[0].view(np.uint32), states[1].view(np.uint32)), "two runs of the same scene differ"
    assert counters[0] == counters[1]
    assert np.isfinite(states[0]).all() and counters[0][3] == 0
