"""sph_ref.py -- independent numpy restatement of the SPH frame step.  TEST INFRASTRUCTURE ONLY.

A second, structurally different statement of Assets/SPHCompute.compute ("compute:line") used
only to cross-check oracle/sph_oracle.cpp (tests/test_golden_oracle.py, tests/golden/make_golden.py) and the golden
fixtures.  It has no grid at all: candidate pairs come from a brute-force N x N test on cell
coordinates (|bin cell of j - unclamped search cell of i| <= 1 per axis, the bin-capacity rule
applied by id rank), so a bug in the oracle's bin bookkeeping cannot hide here.  Neighbour
sums are accumulated in float64 (the oracle sums in fp32 in visiting order), so fields are
compared to a tolerance; neighbour LISTS and all integer artefacts are compared exactly.
N <= ~4096 (dense N x N matrices).

PARITY STATUS: parity unpinned by the reference (it ships no tests or golden vectors).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
PI = f32(3.14159265359)       # compute:20
EPS = f32(1e-6)               # compute:21

POS, VEL, ACC = slice(0, 3), slice(3, 6), slice(6, 9)
DENS, PRES, RHO0, VISC, MASS = 9, 10, 11, 12, 13


class Ref:
    def __init__(self, params, particles):
        """params: sph_amd.abi.SphParams (only read); particles: (N, 18) float32 AoS."""
        self.p = params
        self.P = np.array(particles, dtype=np.float32, copy=True)
        self.n = self.P.shape[0]
        h = f32(params.smoothingRadius)
        self.h, self.h2 = h, f32(h * h)
        h9 = f32(np.power(h, f32(9.0), dtype=np.float32))
        h6 = f32(np.power(h, f32(6.0), dtype=np.float32))
        self.poly6 = f32(f32(315.0) / f32(f32(f32(64.0) * PI) * h9))
        self.spiky = f32(f32(-45.0) / f32(PI * h6))
        self.visc_lap = f32(f32(45.0) / f32(PI * h6))
        self.lap_col = f32(f32(1890.0) / f32(f32(f32(64.0) * PI) * h9))
        self.res = np.array([params.gridResolutionX, params.gridResolutionY, params.gridResolutionZ], dtype=np.int64)
        self.cap = int(params.maxParticlesPerCell)
        self.min_bound = np.array(list(params.minBound), dtype=np.float32)
        self.cell_size = f32(params.cellSize)
        self.collisions = 0
        self.bin_cell = None      # clamped (N,3) cells at the last grid build
        self.bin_rank = None      # rank of each particle inside its bin (ascending id)

    # compute:125 / 155 / 226
    def _cells(self):
        f = ((self.P[:, POS] - self.min_bound[None, :]) / self.cell_size).astype(np.float32)
        return np.floor(f).astype(np.int64)

    # CS_ClearGrid + CS_BuildGrid (compute:104-136), rules O1/O2
    def build_grid(self):
        c = self._cells()
        self.out_of_grid = int(np.any((c < 0) | (c >= self.res[None, :]), axis=1).sum())
        c = np.clip(c, 0, self.res[None, :] - 1)
        self.bin_cell = c
        lin = c[:, 0] + c[:, 1] * self.res[0] + c[:, 2] * self.res[0] * self.res[1]
        self.cell_of_particle = lin.astype(np.uint32)
        order = np.argsort(lin, kind="stable")            # (cell, id)
        rank = np.empty(self.n, dtype=np.int64)
        sl = lin[order]
        starts = np.r_[0, np.flatnonzero(sl[1:] != sl[:-1]) + 1]
        run_start = np.repeat(starts, np.diff(np.r_[starts, self.n]))
        rank[order] = np.arange(self.n) - run_start
        self.bin_rank = rank
        self.sorted_ids = order.astype(np.int32)
        counts = np.bincount(lin, minlength=int(np.prod(self.res)))
        self.grid_counts = counts.astype(np.int32)
        self.bin_overflow_cells = int((counts > self.cap).sum())

    def _pairs(self):
        """Boolean (N,N): j is a CANDIDATE of i; plus rSq (fp32, rule O5 order) and r."""
        search = self._cells()                             # unclamped, rule O7
        d = self.bin_cell[None, :, :] - search[:, None, :]
        cand = np.all(np.abs(d) <= 1, axis=2)
        cand &= (self.bin_rank < self.cap)[None, :]        # rule O2
        np.fill_diagonal(cand, False)                      # compute:173
        x = self.P[:, POS]
        r = (x[:, None, :] - x[None, :, :]).astype(np.float32)
        rsq = ((r[..., 0] * r[..., 0] + r[..., 1] * r[..., 1]).astype(np.float32) + r[..., 2] * r[..., 2]).astype(np.float32)
        return cand, d, r, rsq

    def neighbor_lists(self):
        """Per particle: neighbour ids in visiting order (z, y, x cell offset, then id)."""
        cand, d, _, rsq = self._pairs()
        hit = cand & (rsq < self.h2)
        out = []
        for i in range(self.n):
            js = np.flatnonzero(hit[i])
            key = np.lexsort((js, d[i, js, 0], d[i, js, 1], d[i, js, 2]))
            out.append(js[key].astype(np.int32))
        return out

    # CS_Clear (compute:87-98)
    def clear(self):
        self.P[:, ACC] = 0
        self.P[:, DENS] = self.P[:, RHO0]
        self.P[:, PRES] = 0

    # CS_DensityPressure (compute:142-193)
    def density_pressure(self):
        cand, _, _, rsq = self._pairs()
        hit = cand & (rsq < self.h2)
        diff = (self.h2 - rsq).astype(np.float64)
        w = np.where(hit, self.P[None, :, MASS].astype(np.float64) * float(self.poly6) * diff ** 3, 0.0)
        dens = np.maximum(w.sum(axis=1), self.P[:, RHO0].astype(np.float64)).astype(np.float32)
        self.P[:, DENS] = dens
        rho0 = self.P[:, RHO0]
        cs, gamma = f32(self.p.soundSpeed), f32(self.p.gamma)
        localB = (rho0 * f32(cs * cs) / gamma).astype(np.float32)
        ratio = (dens / rho0).astype(np.float32)
        self.P[:, PRES] = (localB * (np.power(ratio, gamma, dtype=np.float32) - f32(1.0))).astype(np.float32)

    # CS_ForceXSPH (compute:199-303), rule O3 (Jacobi: all reads from the pre-kernel state)
    def force_xsph(self):
        cand, _, r, rsq = self._pairs()
        hit = cand & (rsq < self.h2)
        P = self.P.astype(np.float64)
        rho, pres, mass, visc = P[:, DENS], P[:, PRES], P[:, MASS], P[:, VISC]
        live = self.P[:, DENS] >= EPS
        h, h2 = float(self.h), float(self.h2)
        rs = rsq.astype(np.float64)
        dist = np.sqrt(rs) + 1e-6
        with np.errstate(divide="ignore", invalid="ignore"):
            dirv = r.astype(np.float64) / dist[..., None]
            diff = h2 - rs
            pterm = (pres / rho ** 2)[:, None] + (pres / rho ** 2)[None, :]
            gradW = float(self.spiky) * (diff ** 2)[..., None] * dirv
            Hm = hit[..., None]
            force = np.where(Hm, -(mass[None, :] * pterm)[..., None] * gradW, 0.0).sum(axis=1)
            dv = P[None, :, VEL] - P[:, None, VEL]
            lapv = float(self.visc_lap) * (h - dist)
            avgv = 0.5 * (visc[:, None] + visc[None, :])
            force += np.where(Hm, (avgv * mass[None, :] / rho[None, :] * lapv)[..., None] * dv, 0.0).sum(axis=1)
            cgrad = np.where(Hm, (mass / rho)[None, :, None] * gradW, 0.0).sum(axis=1)
            lapW = -float(self.lap_col) * diff * (h2 - 5.0 * rs)
            clap = np.where(hit, (mass / rho)[None, :] * lapW, 0.0).sum(axis=1)
            W = float(self.poly6) * diff ** 3
            mr = mass[None, :] / (0.5 * (rho[:, None] + rho[None, :]))
            vcorr = np.where(Hm, dv * (W * mr)[..., None], 0.0).sum(axis=1)
            norm = np.where(hit, W * mr, 0.0).sum(axis=1)
        force[:, 1] += float(self.p.gravity) * rho
        gmag = np.linalg.norm(cgrad, axis=1)
        st = gmag > float(self.p.surfaceTensionThreshold)
        with np.errstate(divide="ignore", invalid="ignore"):
            curv = clap / (gmag + 1e-6)
            stf = -float(self.p.surfaceTensionCoefficient) * curv[:, None] * (cgrad / gmag[:, None])
        force = np.where(st[:, None], force + np.nan_to_num(stf), force)
        acc = force / rho[:, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            vnew = P[:, VEL] + np.where((norm > 1e-6)[:, None], vcorr * (float(self.p.xsphEpsilon) / norm)[:, None], 0.0)
        self.P[live, ACC] = acc[live].astype(np.float32)
        self.P[live, VEL] = vnew[live].astype(np.float32)

    # CS_VvHalfStep (compute:309-319) -- elementwise fp32, same operation order as the oracle
    def half_step(self):
        dt = f32(self.p.deltaTime)
        hdt = f32(f32(0.5) * dt)
        self.P[:, VEL] = (self.P[:, VEL] + (hdt * self.P[:, ACC]).astype(np.float32)).astype(np.float32)
        self.P[:, POS] = (self.P[:, POS] + (self.P[:, VEL] * dt).astype(np.float32)).astype(np.float32)

    # CS_VvFullStep (compute:325-334)
    def full_step(self):
        hdt = f32(f32(0.5) * f32(self.p.deltaTime))
        self.P[:, VEL] = (self.P[:, VEL] + (hdt * self.P[:, ACC]).astype(np.float32)).astype(np.float32)

    # CS_BoundObs (compute:340-486).  Elementwise fp32 in the oracle's operation order: the clamp /
    # inside-sphere decisions are discontinuities, so they must see identically rounded values.
    def bound_obs(self):
        p = self.p
        F = np.float32
        x = [self.P[:, k].copy() for k in range(0, 3)]
        v = [self.P[:, k].copy() for k in range(3, 6)]

        def dot3(a, b):
            return ((a[0] * b[0] + a[1] * b[1]).astype(F) + a[2] * b[2]).astype(F)

        he = [F(t) for t in p.boundaryHalfExtents]
        if np.sqrt(F(F(he[0] * he[0] + he[1] * he[1]) + he[2] * he[2])) > F(0.001):
            M = np.array(list(p.boundaryMatrix), dtype=F).reshape(4, 4)
            Mi = np.array(list(p.boundaryInvMatrix), dtype=F).reshape(4, 4)

            def mul4(A, r, vec):
                return ((((A[r, 0] * vec[0] + A[r, 1] * vec[1]).astype(F) + A[r, 2] * vec[2]).astype(F)) + A[r, 3] * F(1.0)).astype(F)

            lp = [mul4(Mi, r, x) for r in range(3)]
            E = F(0.5)
            below = [l < -E for l in lp]
            above = [(l > E) & ~b for l, b in zip(lp, below)]
            collided = below[0] | above[0] | below[1] | above[1] | below[2] | above[2]
            lpc = [np.where(b, -E, np.where(a, E, l)).astype(F) for l, b, a in zip(lp, below, above)]
            ln = [np.where(b, F(-1.0), np.where(a, F(1.0), F(0.0))).astype(F) for b, a in zip(below, above)]
            with np.errstate(divide="ignore", invalid="ignore"):
                l = np.sqrt(dot3(ln, ln))
                ln = [(t / l).astype(F) for t in ln]
                wn = [((M[r, 0] * ln[0] + M[r, 1] * ln[1]).astype(F) + M[r, 2] * ln[2]).astype(F) for r in range(3)]
                wl = np.sqrt(dot3(wn, wn))
                wn = [(t / wl).astype(F) for t in wn]
            xw = [mul4(M, r, lpc) for r in range(3)]
            d = dot3(v, wn)
            for k in range(3):
                vN = (d * wn[k]).astype(F)
                vT = (v[k] - vN).astype(F)
                v[k] = np.where(collided, ((-vN) + vT).astype(F), v[k])
                x[k] = np.where(collided, xw[k], x[k])
            self.collisions += int(collided.sum())
        else:
            for k in range(3):
                hs = F(F(p.boundsSize[k]) * F(0.5))
                mn, mx = F(F(p.boundsCenter[k]) - hs), F(F(p.boundsCenter[k]) + hs)
                lo = x[k] < mn
                hi = (x[k] > mx) & ~lo
                v[k] = np.where(lo & (v[k] < 0), (-v[k] * F(0.9)).astype(F), v[k])
                v[k] = np.where(hi & (v[k] > 0), (-v[k] * F(0.9)).astype(F), v[k])
                x[k] = np.where(lo, mn, np.where(hi, mx, x[k])).astype(F)
        R = F(p.obstacleRadius)
        if R > 0.0:
            damp = F(p.particleCollisionDamping)
            tp = [(x[k] - F(p.obstaclePos[k])).astype(F) for k in range(3)]
            dist = np.sqrt(dot3(tp, tp))
            inside = (dist < R) & (dist > EPS)
            centre = ~inside & (dist <= EPS)
            with np.errstate(divide="ignore", invalid="ignore"):
                nrm = [(t / dist).astype(F) for t in tp]
            pen = (R - dist).astype(F)
            d = dot3(v, nrm)
            for k in range(3):
                x[k] = np.where(inside, (x[k] + (nrm[k] * pen).astype(F)).astype(F), x[k])
                v[k] = np.where(inside, ((v[k] - ((F(2.0) * nrm[k]).astype(F) * d).astype(F)).astype(F) * damp).astype(F), v[k])
            if centre.any():
                x[1] = np.where(centre, (x[1] + F(R + F(0.01))).astype(F), x[1])
                vy = v[1].copy()
                v[0] = np.where(centre, (v[0] * damp).astype(F), v[0])
                v[1] = np.where(centre, ((vy - (F(2.0) * vy).astype(F)).astype(F) * damp).astype(F), v[1])
                v[2] = np.where(centre, (v[2] * damp).astype(F), v[2])
        for k in range(3):
            self.P[:, k] = x[k]
            self.P[:, 3 + k] = v[k]

    # CS_VelocityDamping (compute:492-509)
    def velocity_damping(self):
        v = self.P[:, VEL]
        speed = np.sqrt(((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(np.float32) + v[:, 2] * v[:, 2]).astype(np.float32))
        k = np.where(speed < f32(0.2), f32(0.98), f32(0.999)).astype(np.float32)
        self.P[:, VEL] = (v * k[:, None]).astype(np.float32)

    # SPH.Update body (SPH.cs:201-222)
    def step_frame(self, sub_steps, time_step):
        self.build_grid()
        self.p.deltaTime = float(f32(f32(time_step) / f32(sub_steps)))
        for _ in range(sub_steps):
            self.half_step()
            self.clear()
            self.density_pressure()
            self.force_xsph()
            self.full_step()
            self.bound_obs()
        self.velocity_damping()
