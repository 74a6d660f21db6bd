"""How many waves of the force kernel leave the hit path for the scan path because a particle needs more than
kMaskWords (24) mask words?  BASELINE config 3, frames 5..25 (the driver's timed window) and 20..50.  GPU."""
import sys
import numpy as np
sys.path.insert(0, ".")
from sph_amd import abi, scenes

eng = abi.Engine(0)
sc = scenes.config3(eng, n=1048576)
sc.sph.use_step_frame = True
sc.start()
for f in range(1, 51):
    sc.frame()
    if f not in (1, 3, 5, 8, 12, 16, 20, 25, 35, 50):
        continue
    s = sc.sph
    rx, ry, rz = s.gridResolutionX, s.gridResolutionY, s.gridResolutionZ
    cs, ce, ids, cop = eng.download_grid()
    cnt = (ce - cs).astype(np.int64).reshape(rz, ry, rx)
    P = eng.download_particles()
    c = np.floor((P[:, 0:3] - s.gridMin[None, :]) / np.float32(s.cellSize)).astype(np.int64)
    # row length = the three x-neighbour cells (clipped), per (dy, dz)
    pad = np.zeros((rz + 2, ry + 2, rx + 2), np.int64)
    pad[1:-1, 1:-1, 1:-1] = cnt
    row3 = pad[:, :, :-2] + pad[:, :, 1:-1] + pad[:, :, 2:]          # [z+1, y+1, x] = cells x-1..x+1
    cx, cy, cz = (np.clip(c[:, k], 0, (rx, ry, rz)[k] - 1) for k in range(3))
    nwords = np.zeros(len(P), np.int64)
    cand = np.zeros(len(P), np.int64)
    big = np.zeros(len(P), bool)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            ln = row3[cz + 1 + dz, cy + 1 + dy, cx]
            nw = (ln + 31) // 32
            big |= nw > 8
            nwords += nw
            cand += ln
    over = (nwords > 24) | big
    order = ids.astype(np.int64)                    # slot -> id
    ov = over[order]
    waves = ov[: len(ov) // 64 * 64].reshape(-1, 64).any(axis=1)
    print(f"frame {f}: candidates/particle {cand.mean():.0f}, mask words mean {nwords.mean():.1f} p99 {np.percentile(nwords, 99):.0f} max {nwords.max()}, "
          f"particles over 24 words {over.mean() * 100:.2f} %, waves on the scan path {waves.mean() * 100:.2f} %", flush=True)
eng.destroy()
