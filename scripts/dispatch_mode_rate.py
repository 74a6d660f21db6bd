"""Frame time of config 3 when the host dispatches kernel by kernel the way SPH.cs does (sph_dispatch per
#pragma kernel, CS_ClearGrid and CS_BuildGrid included) against sph_step_frame.  Wall clock over K frames.
  python scripts/dispatch_mode_rate.py [warmup=5] [frames=20] [n=1048576]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import abi, scenes

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1048576
for fused in (True, False):
    e = abi.Engine(0)
    sc = scenes.config3(e, n=n)
    sc.sph.use_step_frame = fused
    sc.start()
    for _ in range(W):
        sc.frame()
    e.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        sc.frame()
    e.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{'sph_step_frame' if fused else 'kernel-by-kernel dispatch'}: {dt * 1e3:.4f} ms per frame, {n / dt / 1e9:.3f} G updates/s")
    e.destroy()
