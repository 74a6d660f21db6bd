"""Long free run of config 3 at full size on the engine alone: frames per second over time, counters, finiteness.
  python scripts/long_run.py [frames=3000] [n=1048576]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import abi, scenes

F = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
e = abi.Engine(0)
sc = scenes.config3(e, n=n)
sc.sph.use_step_frame = True
sc.start()
t0 = time.perf_counter()
for f in range(1, F + 1):
    sc.frame()
    if f % 500 == 0:
        e.synchronize()
        dt = time.perf_counter() - t0
        p = e.download_particles()
        print(f"frame {f}: {dt / 500 * 1e3:.3f} ms/frame over the last 500; finite {bool(np.isfinite(p).all())}; "
              f"overflow cells {e.read_counter(abi.CTR_BIN_OVERFLOW)}, out of grid {e.read_counter(abi.CTR_OUT_OF_GRID)}, "
              f"non-finite {e.read_counter(abi.CTR_NAN)}; y range {p[:, 1].min():.2f}..{p[:, 1].max():.2f}, "
              f"max |v| {np.abs(p[:, 3:6]).max():.2f}, density {p[:, 9].min():.1f}..{p[:, 9].max():.1f}", flush=True)
        t0 = time.perf_counter()
e.destroy()
