"""Raw-state files and the device render feed through the C ABI.  -m gpu."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libsph_hip is loaded: torch ships its own HIP runtime, and two in one process do not share the GPU)

from sph_amd import abi, scenes, statefile
from tests.test_statefile import _resume_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step_frame", [False, True])
def test_resume_is_bitwise_engine(tmp_path, step_frame):
    def builder(e, **kw):
        sc = scenes.config3(e, **kw)
        sc.sph.use_step_frame = step_frame
        return sc
    _resume_case(lambda: abi.Engine(0), tmp_path, builder, 4, 4, n=40000, scale=0.55)


def test_state_saved_by_engine_loads_into_oracle(tmp_path):
    """The file carries everything the next frame needs: the oracle continues an engine run within tolerance."""
    from oracle.ora import Oracle
    from tests.util import assert_fields_close
    eng = abi.Engine(0)
    se = scenes.unity_scene(eng, which=4, scale_count=0.02)
    se.start(); se.run(3)
    prefix = str(tmp_path / "e")
    statefile.save_state(se, prefix)
    se.run(1)
    ora = Oracle()
    so = scenes.unity_scene(ora, which=4, scale_count=0.02)
    statefile.load_state(so, prefix)
    so.run(1)
    assert_fields_close(eng.download_particles(), ora.download_particles(), 1e-5, "frame after reload: ")


def test_device_render_feed_equals_download():
    import torch
    eng = abi.Engine(0)
    sc = scenes.unity_scene(eng, which=2, scale_count=0.02)
    sc.sph.use_step_frame = True
    sc.start(); sc.run(3)
    n = eng.n
    feed = torch.full((n, 18), float("nan"), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    eng.export_particles_device(feed.data_ptr())
    eng.synchronize()
    host = eng.download_particles()
    assert np.array_equal(feed.cpu().numpy().view(np.uint32), host.view(np.uint32))
    # wrong count, a pointer past the end of its allocation, a host pointer: refused, nothing launched
    # (the size check is against the extent of the hipMalloc allocation the pointer lies in)
    with pytest.raises(abi.SphError):
        eng.export_particles_device(feed.data_ptr(), n=n - 1)
    base, size = eng.device_pointer("acc")
    with pytest.raises(abi.SphError):
        eng.export_particles_device(base + size - 64)
    with pytest.raises(abi.SphError):
        eng.export_particles_device(host.ctypes.data)
    eng.destroy()
