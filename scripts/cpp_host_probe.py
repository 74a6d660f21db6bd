"""C++ host vs Python host on BASELINE config 3 (rocking cube + orbiting sphere): how far apart after F frames?"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sph_amd import abi, scenes
frames, scale = int(sys.argv[1]), float(sys.argv[2])
out = subprocess.run([os.path.join(ROOT, "host", "sph_headless"), "--config", "3", "--scale", str(scale), "--frames", str(frames),
                      "--step-frame", "--dump", "/tmp/cpp3.bin"], check=True, capture_output=True, text=True).stdout
info = json.loads(out.strip().splitlines()[-1]); print(info)
got = np.fromfile("/tmp/cpp3.bin", dtype=np.float32).reshape(-1, 18)
eng = abi.Engine(0)
sc = scenes.config3(eng, n=info["particles"])
sc.sph.use_step_frame = True
sc.run(frames)
want = eng.download_particles()
print("python grid", sc.sph.gridResolutionX, sc.sph.gridResolutionY, sc.sph.gridResolutionZ)
for name, sl in (("pos", slice(0, 3)), ("vel", slice(3, 6)), ("acc", slice(6, 9)), ("rho", slice(9, 10)), ("p", slice(10, 11))):
    d = np.abs(got[:, sl].astype(np.float64) - want[:, sl]).max(); s = np.abs(want[:, sl]).max()
    print(name, "max abs diff", d, "scale", s, "rel", d / s, "identical", np.array_equal(got[:, sl], want[:, sl]))
