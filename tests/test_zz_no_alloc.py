"""sph_slab_plane_counts_noalloc -- the one entry point that allocates nothing (static device scratch) -- against
sph_slab_plane_counts, through the C ABI: bit-exact (integer counts), two calls in a row, the second with a smaller
input; the scratch keeps the first call's contents in between.  Collected last (test_zz_*): the round's end runs
pytest -x.  -m gpu."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libsph_hip is loaded: torch ships its own HIP runtime, and two in one process do not share the GPU)

from sph_amd import abi

pytestmark = pytest.mark.gpu


def _slab_world1(n, cube, spawn, grid_resolution):
    """One rank in slab mode (both entry points are available there) over its own tank."""
    import torch
    from sph_amd import scenes, slab
    from sph_amd.host import SPH, Scene, SpawnBox, Time, Transform
    eng = abi.Engine(0)
    se = slab.SlabEngine(slab.DevicePort(eng, 0, msg_capacity=4096), slab.NeighbourComm(0, 1, torch.device("cuda", 0), backend="gloo"), 0, 1)
    f = dict(scenes.SCENE_COMMON)
    f.update(gridResolution=grid_resolution)
    tank = Transform(position=(0, 0.5 * cube[1], 0), localScale=cube)
    boxes = [SpawnBox(n, spawn, 20, 0.1, 0.2, position=(0.3, 0.5 * cube[1], 0.0), initialVelocity=(1.0, 0, 0))]
    sc = Scene(SPH(se, **f, surfaceTensionCoefficient=0.1, xsphEpsilon=0.3, boundaryCube=tank, spawnBoxes=boxes), Time(), [])
    sc.sph.use_step_frame = True
    sc.start()
    sc.run(3)
    return eng, se


def test_noalloc_plane_counts_equal_the_allocating_entry_point():
    import torch
    import torch.distributed as dist
    from tests.test_slab_gloo import _free_port
    dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{_free_port()}")
    try:
        # first call: 96 planes; second call, right after it: 40 planes (a smaller input over the same scratch)
        big, se_big = _slab_world1(6000, (24.0, 4.0, 2.0), (16.0, 1.0, 1.0), 8)
        small, se_small = _slab_world1(1500, (10.0, 3.0, 2.0), (5.0, 0.8, 0.8), 8)
        rx_big, rx_small = se_big.grid[0], se_small.grid[0]
        assert rx_big == 96 and rx_small == 40
        want_big = big.slab_plane_counts(rx_big)
        want_small = small.slab_plane_counts(rx_small)
        assert want_big.sum() == big.particle_count() == 6000 and want_small.sum() == 1500
        assert np.count_nonzero(want_big) > 40 and np.count_nonzero(want_big[rx_small:]) > 0   # stale counts WOULD show
        got_big = big.slab_plane_counts_noalloc(rx_big)
        got_small = small.slab_plane_counts_noalloc(rx_small)
        assert np.array_equal(got_big, want_big), "first call differs from sph_slab_plane_counts"
        assert np.array_equal(got_small, want_small), "second (smaller) call differs from sph_slab_plane_counts"
        # and again in the other order, into DEVICE memory of the caller
        dev = torch.full((rx_big,), -1, dtype=torch.int32, device="cuda:0")
        small.slab_plane_counts_noalloc(rx_small, out=dev.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(dev[:rx_small].cpu().numpy(), want_small) and int(dev[rx_small]) == -1
        big.slab_plane_counts_noalloc(rx_big, out=dev.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(dev.cpu().numpy(), want_big)
        # sizes beyond the static scratch are refused with a clear error, as is a count that is not the grid's
        with pytest.raises(abi.SphError, match="static scratch"):
            big.slab_plane_counts_noalloc(abi.PLANE_COUNTS_MAX + 1)
        with pytest.raises(abi.SphError, match="plane counts"):
            big.slab_plane_counts_noalloc(rx_big - 1)
        big.destroy()
        small.destroy()
    finally:
        dist.destroy_process_group()


def test_noalloc_plane_counts_outside_slab_mode():
    """A plain one-GPU engine: planes of the whole grid, against the cell ranges of sph_download_grid."""
    from sph_amd import scenes
    eng = abi.Engine(0)
    sc = scenes.unity_scene(eng, which=2, scale_count=0.02)
    sc.sph.use_step_frame = True
    sc.run(2)
    rx = sc.sph.gridResolutionX
    cs, ce, _, _ = eng.download_grid()
    want = (ce - cs).astype(np.int64).reshape(-1, rx).sum(axis=0)      # reference cell order: x fastest (compute:127)
    got = eng.slab_plane_counts_noalloc(rx)
    assert np.array_equal(got, want) and got.sum() == eng.n
    eng.destroy()
