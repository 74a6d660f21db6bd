"""Is the frame loop host-bound?  Time to ENQUEUE K frames (no synchronisation) against the time the GPU needs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import abi, scenes
eng = abi.Engine(0)
sc = scenes.config3(eng)
sc.sph.use_step_frame = True
sc.start()
for _ in range(20): sc.frame()
eng.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K): sc.frame()
t1 = time.perf_counter()
eng.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.4f} ms/frame, total {1e3*(t2-t0)/K:.4f} ms/frame")
# the same with the host work outside sph_step_frame removed
t0 = time.perf_counter()
for _ in range(K): eng.step_frame(1, 0.02)
t1 = time.perf_counter()
eng.synchronize()
t2 = time.perf_counter()
print(f"step_frame only: enqueue {1e3*(t1-t0)/K:.4f} ms/frame, total {1e3*(t2-t0)/K:.4f} ms/frame")
