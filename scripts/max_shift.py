"""Largest plane shift per frame (SPH_CTR_MAX_PLANE_SHIFT) of a workload, on the slab code path with one rank:
what ghost width 2 * subSteps * (1 + shift) the workload needs.  usage: python scripts/max_shift.py [particles] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from sph_amd import abi, scenes, slab
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8388608
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 80
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("gloo", rank=0, world_size=1)
eng = abi.Engine(0)
se = slab.SlabEngine(slab.DevicePort(eng, 0, msg_capacity=65536), slab.NeighbourComm(0, 1, torch.device("cuda", 0), backend="gloo"), 0, 1)
sc = scenes.config5(se, gpus=1, per_gpu=n)
sc.sph.use_step_frame = True
sc.start()
shifts = []
for f in range(frames):
    sc.frame()
    shifts.append(eng.read_counter(abi.CTR_MAX_PLANE_SHIFT))
print("max plane shift per frame:", shifts)
print("largest", max(shifts), "-> ghost planes needed", 2 * (1 + max(shifts)))
