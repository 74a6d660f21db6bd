"""How much of the neighbour kernels' lane time is useful?  Runs config 3 for W frames, then from the
exported grid and positions computes, per wave of 64 consecutive slots: candidate groups per lane vs
the per-row maximum the wave actually executes (density kernel), and hits per lane vs the wave
maximum (force kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_amd import abi, scenes

W = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
eng = abi.Engine(0)
wl = sys.argv[3] if len(sys.argv) > 3 else "config3"
sc = scenes.config3(eng, n=n) if wl == "config3" else scenes.config5(eng, gpus=1, per_gpu=n)
sc.sph.use_step_frame = True
sc.start()
for _ in range(W):
    sc.frame()
P = eng.download_particles()
cs, ce, ids, cop = eng.download_grid()
cnt, _ = eng.download_neighbors(0)
sph = sc.sph
rx, ry, rz = sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ
prm = sph.build_params()
mn = np.array(prm.minBound[:3], dtype=np.float32)
cell = np.float32(prm.cellSize)
# NOTE: the kernels search around the cell of the position at density time (after the half step); the
# end-of-frame positions used here differ by the full-step/boundary correction only -- fine for statistics
c = np.floor((P[:, 0:3] - mn) / cell).astype(np.int64)
cum = np.concatenate([[0], ce]).astype(np.int64)         # reference order: cum[c] = start of cell c
order = ids                                              # slot -> id  (tiled memory order differs; statistics only)
pid = order
cx, cy, cz = c[pid, 0], c[pid, 1], c[pid, 2]
groups = np.zeros((9, len(pid)), dtype=np.int64)
cands = np.zeros(len(pid), dtype=np.int64)
t = 0
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        y, z = cy + dy, cz + dz
        ok = (y >= 0) & (y < ry) & (z >= 0) & (z < rz)
        xlo, xhi = np.clip(cx - 1, 0, rx - 1), np.clip(cx + 1, 0, rx - 1)
        base = (np.clip(y, 0, ry - 1) + ry * np.clip(z, 0, rz - 1)) * rx
        ln = np.where(ok, cum[base + xhi + 1] - cum[base + xlo], 0)
        groups[t] = (ln + 3) // 4
        cands += ln
        t += 1
hits = cnt[pid]
m = len(pid) // 64 * 64
g = groups[:, :m].reshape(9, -1, 64)
lane_groups = g.sum(0)                                  # per lane
wave_groups = g.max(2).sum(0)                           # what the wave executes (per-row maximum)
flat_groups = lane_groups.max(1)                        # if rows were flattened per lane
h = hits[:m].reshape(-1, 64)
print(f"frames {W} particles {len(pid)}")
print(f"candidates/particle mean {cands.mean():.1f}  hits/particle mean {hits.mean():.1f}  hit fraction {hits.sum()/cands.sum():.3f}")
print(f"density: groups/lane mean {lane_groups.mean():.1f}; wave executes {wave_groups.mean():.1f} (efficiency {lane_groups.mean()/wave_groups.mean():.2f}); "
      f"flattened would execute {flat_groups.mean():.1f} (efficiency {lane_groups.mean()/flat_groups.mean():.2f})")
print(f"force: hits/lane mean {h.mean():.1f}; wave executes max {h.max(1).mean():.1f} (efficiency {h.mean()/h.max(1).mean():.2f})")
hs = np.sort(h, axis=1)
print(f"force: if lanes were sorted by hits in blocks of 256 -> wave max mean {np.sort(hits[:len(hits)//256*256].reshape(-1,256),axis=1).reshape(-1,4,64).max(2).mean():.1f}")
words = ((groups * 4 + 31) // 32)          # upper estimate of mask words per row (groups are 4-candidate units)
wtot = np.zeros(len(pid), dtype=np.int64)
t = 0
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        y, z = cy + dy, cz + dz
        ok = (y >= 0) & (y < ry) & (z >= 0) & (z < rz)
        xlo, xhi = np.clip(cx - 1, 0, rx - 1), np.clip(cx + 1, 0, rx - 1)
        base = (np.clip(y, 0, ry - 1) + ry * np.clip(z, 0, rz - 1)) * rx
        ln = np.where(ok, cum[base + xhi + 1] - cum[base + xlo], 0)
        wtot += (ln + 31) // 32
mw = wtot[:m].reshape(-1, 64).max(1)
print(f"mask words/particle mean {wtot.mean():.1f} max {wtot.max()}; particles > 16 words: {(wtot > 16).mean()*100:.2f} %; waves with such a lane: {(mw > 16).mean()*100:.1f} %; > 24: {(mw > 24).mean()*100:.1f} %")
