"""Design statistics for a force kernel whose quads (4 adjacent lanes) walk their hit lists in lock step by slot, so
that the four lanes of a quad gather the SAME neighbour record per instruction (1 vector-L1 access per quad instead of
one per distinct address).  CPU only, on the OpenMP oracle's state of config 3 after W frames:
  * trips per wave today  = max over the 64 lanes of the lane's hit count,
  * trips per wave merged = max over the 16 quads of |union of the four lanes' hit sets|,
  * distinct 64-byte blocks a quad touches per gather today (its four lanes' k-th hits) against 1 merged.
Usage: python scripts/quad_union_stats.py [frames=8] [n=1048576] [sample_waves=4000] [group=4]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ora import Oracle
from sph_amd import abi, scenes

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
SAMPLE = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
G = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ora = Oracle(omp=True)
sc = scenes.config3(ora, n=n)
sc.sph.use_step_frame = True
sc.start()
for _ in range(W):
    sc.frame()
sph = sc.sph
sph.use_step_frame = False
for m in sc.movers:
    m.Update()
if sph.boundaryCube is not None:
    sph.UpdateGridParametersFromBoundary()
sph.UpdateObstacle()
sph.SetComputeParams()
sph.UpdateParticleStaticProperties()
sph.DispatchCompute(abi.K_CLEAR_GRID, sph.totalCells)
sph.DispatchCompute(abi.K_BUILD_GRID, sph.particleCount)
ora.set_delta_time(np.float32(sph.timeStep / sph.subSteps))
cs, ce, ids, cop = ora.download_grid()
sph.DispatchCompute(abi.K_VV_HALF_STEP, sph.particleCount)
sph.DispatchCompute(abi.K_DENSITY_PRESSURE, sph.particleCount)
counts, lists = ora.download_neighbors(256)
rx, ry, rz = sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ
TY = TZ = 8
ryT = (ry + TY - 1) // TY
bc = cop.astype(np.int64)
bx, by, bz = bc % rx, (bc // rx) % ry, bc // (rx * ry)
key = bx + rx * ((((bz // TZ) * ryT + (by // TY)) * (TY * TZ)) + (bz % TZ) * TY + (by % TY))
order = np.lexsort((np.arange(n), key))            # slot -> id
slot_of = np.empty(n, dtype=np.int64)
slot_of[order] = np.arange(n)
nw = n // 64
rs = np.random.RandomState(2)
waves = rs.choice(nw, size=min(SAMPLE, nw), replace=False)
now, merged, hits_tot, union_tot, blocks_now = [], [], 0, 0, []
for w in waves:
    idw = order[w * 64:(w + 1) * 64]
    c = counts[idw]
    now.append(c.max())
    hits_tot += c.sum()
    mx = 0
    for q in range(64 // G):
        lanes = idw[q * G:(q + 1) * G]
        sets = [slot_of[lists[i, :counts[i]]] for i in lanes]
        u = np.unique(np.concatenate(sets)).size if sum(s.size for s in sets) else 0
        union_tot += u
        mx = max(mx, u)
        # today: at step k the lanes gather their own k-th hit; 16-byte records, 64-byte blocks
        kmax = max(s.size for s in sets)
        if kmax:
            for k in range(0, kmax, 7):
                a = [s[k] >> 2 for s in sets if s.size > k]
                blocks_now.append(len(set(a)))
    merged.append(mx)
now, merged = np.array(now), np.array(merged)
print(f"frame {W}, {n} particles, lanes per group {G}: hits per lane mean {hits_tot / (len(waves) * 64):.1f}")
print(f"trips per wave today (max lane hit count): mean {now.mean():.1f};  merged (max group union): mean {merged.mean():.1f}  -> x{merged.mean() / now.mean():.2f}")
print(f"pair evaluations per wave-trip (lane activity): today {hits_tot / (now.sum() * 64):.2f}, merged {hits_tot / (merged.sum() * 64):.2f}")
print(f"mean union per group {union_tot / (len(waves) * 64 / G):.1f};  distinct 64-byte blocks per group and gather today: {np.mean(blocks_now):.2f} (merged: 1)")
