import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import abi, scenes
a = abi.Engine(0)
os.environ["SPH_HIP_HITMASK"] = "0"
b = abi.Engine(0)
sa, sb = scenes.unity_scene(a, which=4, scale_count=0.05), scenes.unity_scene(b, which=4, scale_count=0.05)
for s in (sa, sb):
    s.sph.use_step_frame = False
    s.start()
names = ["px","py","pz","vx","vy","vz","ax","ay","az","rho","p","rho0","visc","mass","c0","c1","c2","c3"]
for f in range(3):
    sa.frame(); sb.frame()
    pa, pb = a.download_particles(), b.download_particles()
    d = pa.view(np.uint32) != pb.view(np.uint32)
    print("frame", f, "rows differing", d.any(axis=1).sum(), "of", len(pa))
    for k in range(18):
        if d[:, k].any():
            i = np.flatnonzero(d[:, k])
            print("  ", names[k], len(i), "max abs diff", np.abs(pa[i, k] - pb[i, k]).max(), "first", i[:5], pa[i[:3], k], pb[i[:3], k])
    if d.any():
        break
