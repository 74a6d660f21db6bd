"""One-rank slab run (SlabEngine, world 1) of config 5 at a given size: stage times of the slab code path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from sph_amd import abi, scenes, slab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
eng = abi.Engine(0)
port = slab.DevicePort(eng, 0, msg_capacity=max(n // 4, 65536))
comm = slab.NeighbourComm(0, 1, torch.device("cuda", 0), backend="gloo")
se = slab.SlabEngine(port, comm, 0, 1, ghost_planes=int(os.environ.get("GHOST", "8")), overlap=os.environ.get("OVERLAP", "0") == "1")
sc = scenes.config5(se, gpus=1, per_gpu=n)
sc.sph.use_step_frame = True
sc.start()
for _ in range(20): sc.frame()
eng.synchronize(); eng.get_timings()
t0 = time.perf_counter()
for _ in range(10): sc.frame()
eng.synchronize()
print("SLAB1", n, round((time.perf_counter() - t0) / 10 * 1e3, 4), {k: round(v, 4) for k, v in eng.get_timings().items() if k.endswith("_ms")})

# where does the host time of one frame go?  (world 1: no messages, only pack / rebuild / launch)
import time as _t
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = _t.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + _t.perf_counter() - t; return r
    return w
port.pack = timed("pack", port.pack); port.rebuild = timed("rebuild", port.rebuild); port.pack_wait = timed("pack_wait", port.pack_wait)
comm.exchange = timed("comm", comm.exchange); se.e.step_frame = timed("step_frame(launch)", se.e.step_frame)
se.exchange = timed("exchange(total)", se.exchange)
sc.sph.SetComputeParams = timed("SetComputeParams", sc.sph.SetComputeParams)
sc.sph.UpdateParticleStaticProperties = timed("UpdateStatic", sc.sph.UpdateParticleStaticProperties)
t0 = _t.perf_counter()
for _ in range(20): sc.frame()
eng.synchronize()
tot = (_t.perf_counter() - t0) / 20 * 1e3
print("HOST ms/frame", round(tot, 4), {k: round(v / 20 * 1e3, 4) for k, v in acc.items()})
