"""Design statistics for per-wave LDS candidate windows (round 3).  CPU only: runs config 3 on the OpenMP oracle for
W frames, then performs grid build + half step of the next frame and looks at what each wave of 64 consecutive slots
(engine order: tiled key, id) would have to stage:

  * how many distinct CURRENT (y, z) rows its lanes sit in (the grid is stale: binned before the half step),
  * with windows = [cells X0-2 .. X1+2] of each distinct row's neighbour rows: slots staged per step, lanes covered,
  * coalesced 64-slot loads needed vs. the per-lane gather groups the kernel issues today.

Usage: python scripts/wave_window_stats.py [frames=8] [n=1048576] [sample_waves=3000]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ora import Oracle
from sph_amd import abi, scenes

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
SAMPLE = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
KMAX = int(os.environ.get("KMAX", 6))
MARGIN = int(os.environ.get("MARGIN", 2))
BUDGET = int(os.environ.get("BUDGET", 384))      # slots of LDS per wave and step

LOAD = os.environ.get("LOAD")
ora = Oracle(omp=True)
sc = scenes.config3(ora, n=n)
sc.sph.use_step_frame = True
sc.start()
for _ in range(W):
    sc.frame()
# next frame, kernel by kernel, up to the half step
sph = sc.sph
sc.sph.use_step_frame = False
for m in sc.movers:
    m.Update()
# replicate SPH.Update up to the half step
e = ora
if sph.boundaryCube is not None:
    sph.UpdateGridParametersFromBoundary()
sph.UpdateObstacle()
sph.SetComputeParams()
sph.UpdateParticleStaticProperties()
sph.DispatchCompute(abi.K_CLEAR_GRID, sph.totalCells)
sph.DispatchCompute(abi.K_BUILD_GRID, sph.particleCount)
e.set_delta_time(np.float32(sph.timeStep / sph.subSteps))
cs, ce, ids, cop = e.download_grid()
sph.DispatchCompute(abi.K_VV_HALF_STEP, sph.particleCount)
P = e.download_particles()

rx, ry, rz = sph.gridResolutionX, sph.gridResolutionY, sph.gridResolutionZ
prm = sph.build_params()
mn = np.array(prm.minBound[:3], dtype=np.float32)
cell = np.float32(prm.cellSize)
TY = TZ = 8
ryT = (ry + TY - 1) // TY


def row_index(y, z):
    return (((z // TZ) * ryT + (y // TY)) * (TY * TZ)) + (z % TZ) * TY + (y % TY)


# build cell of every particle (reference linear index) -> tiled key
bc = cop.astype(np.int64)
bx, by, bz = bc % rx, (bc // rx) % ry, bc // (rx * ry)
key = bx + rx * row_index(by, bz)
order = np.lexsort((np.arange(n), key))             # slot -> id
skey = key[order]
nrows_t = ryT * TY * ((rz + TZ - 1) // TZ) * TZ
cum = np.zeros(rx * nrows_t + 1, dtype=np.int64)
np.add.at(cum, skey + 1, 1)
cum = np.cumsum(cum)
# current (search) cell after the half step
c = np.floor((P[:, 0:3] - mn) / cell).astype(np.int64)
c[:, 0] = np.clip(c[:, 0], -2, rx + 1); c[:, 1] = np.clip(c[:, 1], -2, ry + 1); c[:, 2] = np.clip(c[:, 2], -2, rz + 1)
cxs, cys, czs = c[order, 0], c[order, 1], c[order, 2]
bxs, bys, bzs = bx[order], by[order], bz[order]
LOCALSORT = int(os.environ.get("LOCALSORT", 0))     # re-sort every block of this many slots by CURRENT cell
if LOCALSORT:
    ck = cxs + 1000 * row_index(np.clip(cys, 0, ry - 1), np.clip(czs, 0, rz - 1))
    blk = np.arange(n) // LOCALSORT
    o2 = np.lexsort((np.arange(n), ck, blk))
    cxs, cys, czs, bxs, bys, bzs = cxs[o2], cys[o2], czs[o2], bxs[o2], bys[o2], bzs[o2]
moved = (cxs != bxs) | (cys != bys) | (czs != bzs)
movedyz = (cys != bys) | (czs != bzs)
print(f"frame {W}: {n} particles, grid {rx}x{ry}x{rz}; moved cell since binning: {moved.mean()*100:.1f} %, moved row (y or z): {movedyz.mean()*100:.1f} %")


def rng_of(y, z, xlo, xhi):
    if y < 0 or y >= ry or z < 0 or z >= rz:
        return 0, 0
    xlo, xhi = max(xlo, 0), min(xhi, rx - 1)
    if xlo > xhi:
        return 0, 0
    b = rx * row_index(y, z)
    return int(cum[b + xlo]), int(cum[b + xhi + 1])


if os.environ.get("SAVE"):
    np.savez(os.environ["SAVE"], bx=bx, by=by, bz=bz, cx=c[:, 0], cy=c[:, 1], cz=c[:, 2], rx=rx, ry=ry, rz=rz)
    print("saved", os.environ["SAVE"])
    sys.exit(0)
nw = n // 64
rs = np.random.RandomState(1)
waves = rs.choice(nw, size=min(SAMPLE, nw), replace=False)
distinct = []
tot_groups_now = 0          # per-lane 4-candidate groups the wave executes today (per-step maxima)
tot_loads_now = 0
tot_chunks = 0              # 64-slot coalesced loads with windows
tot_steps = 0
uncovered_lane_steps = 0
lane_steps = 0
steps_with_uncovered = 0
stage_slots = []
groups_serial_extra = 0
for w in waves:
    sl = slice(w * 64, w * 64 + 64)
    X, Y, Z = cxs[sl], cys[sl], czs[sl]
    rowid = Y * 100000 + Z
    # distinct rows by first occurrence
    groups = []
    seen = {}
    for l in range(64):
        r = rowid[l]
        if r not in seen:
            seen[r] = len(groups)
            groups.append([l])
        else:
            groups[seen[r]].append(l)
    distinct.append(len(groups))
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            tot_steps += 1
            # per lane ranges
            a = np.zeros(64, dtype=np.int64); en = np.zeros(64, dtype=np.int64)
            for l in range(64):
                a[l], en[l] = rng_of(Y[l] + dy, Z[l] + dz, X[l] - 1, X[l] + 1)
            g = (en - a + 3) // 4
            tot_groups_now += g.max()
            tot_loads_now += g.max() * 3
            used = 0
            covered = np.zeros(64, dtype=bool)
            for k, lanes in enumerate(groups):
                if k >= KMAX:
                    break
                if LOCALSORT or os.environ.get('EXACTX'):
                    x0 = X[lanes].min() - 1; x1 = X[lanes].max() + 1
                else:
                    x0 = X[lanes[0]] - MARGIN; x1 = X[lanes[-1]] + MARGIN
                    if x1 < x0:
                        x0, x1 = x1 - 2 * MARGIN + 0, x0 + 2 * MARGIN
                A, E = rng_of(Y[lanes[0]] + dy, Z[lanes[0]] + dz, x0, x1)
                ch = (E - A + 63) // 64
                if used + ch * 64 > BUDGET:
                    continue
                used += ch * 64
                tot_chunks += ch
                for l in lanes:
                    if a[l] >= A and en[l] <= E:
                        covered[l] = True
            empty = en <= a
            unc = ~covered & ~empty
            uncovered_lane_steps += unc.sum()
            lane_steps += (~empty).sum()
            stage_slots.append(used)
            if unc.any():
                steps_with_uncovered += 1
                groups_serial_extra += g[unc].max()
distinct = np.array(distinct)
print(f"distinct current rows per wave: mean {distinct.mean():.2f}, hist {np.bincount(distinct)[:12]}")
print(f"today: {tot_groups_now/len(waves):.1f} 4-candidate groups per wave (sum of per-step maxima), {tot_loads_now/len(waves):.0f} gather loads per wave")
print(f"windows (K<={KMAX}, margin {MARGIN}, budget {BUDGET} slots): {tot_chunks/len(waves):.1f} coalesced 64-slot loads per wave, "
      f"staged slots per step mean {np.mean(stage_slots):.0f} p95 {np.percentile(stage_slots,95):.0f} max {np.max(stage_slots)}")
print(f"lane-steps uncovered: {uncovered_lane_steps/lane_steps*100:.2f} %; steps with an uncovered lane: {steps_with_uncovered/tot_steps*100:.1f} %; "
      f"serial second-pass groups per wave: {groups_serial_extra/len(waves):.1f} (+{groups_serial_extra/tot_groups_now*100:.1f} % trips)")
