"""The C++ headless host (host/sph_headless, the compiled-language mirror of Assets/SPH.cs) and the
Python host (sph_amd.host) drive the same C ABI and must produce the same particles."""
import json
import os
import subprocess

import numpy as np
import pytest

from sph_amd import abi, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "host", "sph_headless")


@pytest.mark.parametrize("step_frame", [False, True])
def test_cpp_host_matches_python_host(tmp_path, step_frame):
    if not os.path.exists(EXE):
        pytest.fail("host/sph_headless is not built (python -c 'import __graft_entry__ as g; g.build()')")
    frames, scale = 6, 0.02
    dump = tmp_path / "cpp.bin"
    cmd = [EXE, "--scene", "4", "--scale", str(scale), "--frames", str(frames), "--dump", str(dump)]
    if step_frame:
        cmd.append("--step-frame")
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=120).stdout
    info = json.loads(out.strip().splitlines()[-1])
    got = np.fromfile(dump, dtype=np.float32).reshape(-1, 18)

    eng = abi.Engine(0)
    sc = scenes.unity_scene(eng, which=4, scale_count=scale)
    sc.sph.use_step_frame = step_frame
    sc.run(frames)
    want = eng.download_particles()
    eng.destroy()
    assert info["particles"] == want.shape[0] and info["grid"] == [80, 120, 80]
    assert np.array_equal(got, want), "C++ host and Python host disagree"


def test_cpp_host_matches_python_host_on_config3_rocking_cube(tmp_path):
    """BASELINE config 3 through both hosts: the cube rocks (regrid every frame, SPH.cs:176-185, 303-346), the sphere
    orbits (SphereMover.cs:12-21); 140 frames at 2 % of the particle count, long enough for the fluid to fall onto the
    rotated walls (OBB collisions, compute:349-419).  Grid sizing, quaternion, TRS matrix, its inverse and the movers are
    computed independently in C++ and in Python: one differing bit in a matrix parts the two runs at the first collision,
    so bit-identical particles at the end say the two hosts hand the kernels the same parameters every frame."""
    if not os.path.exists(EXE):
        pytest.fail("host/sph_headless is not built (python -c 'import __graft_entry__ as g; g.build()')")
    frames, scale = 140, 0.02
    dump = tmp_path / "cpp3.bin"
    out = subprocess.run([EXE, "--config", "3", "--scale", str(scale), "--frames", str(frames), "--step-frame", "--dump", str(dump)],
                         check=True, capture_output=True, text=True, timeout=120).stdout
    info = json.loads(out.strip().splitlines()[-1])
    got = np.fromfile(dump, dtype=np.float32).reshape(-1, 18)
    eng = abi.Engine(0)
    sc = scenes.config3(eng, n=info["particles"])
    sc.sph.use_step_frame = True
    sc.run(frames)
    want = eng.download_particles()
    assert info["grid"] == [sc.sph.gridResolutionX, sc.sph.gridResolutionY, sc.sph.gridResolutionZ] != [128, 192, 128]
    assert info["collisions_last_frame"] == eng.read_counter(abi.CTR_COLLISIONS) > 0
    eng.destroy()
    assert np.array_equal(got, want), "C++ host and Python host disagree on config 3"
