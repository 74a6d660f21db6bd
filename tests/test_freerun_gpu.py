"""Free-run regime: BASELINE config 1 (16,384 particles, single-phase dam break in the
AABB box) for 1000 frames on the engine and on the oracle, each from the same initial state and never
re-synchronised.  SPH is chaotic and rSq < h2 is a discontinuity, so no per-particle claim is made:
aggregate quantities must stay within the stated drift bounds."""
import numpy as np
import pytest

from sph_amd import abi, scenes

pytestmark = pytest.mark.gpu


def aggregates(P):
    v = P[:, 3:6].astype(np.float64)
    m = P[:, 13].astype(np.float64)
    return {
        "com": (P[:, 0:3].astype(np.float64) * m[:, None]).sum(0) / m.sum(),
        "ke": float((0.5 * m * (v ** 2).sum(1)).sum()),
        "pe": float((m * 9.81 * P[:, 1].astype(np.float64)).sum()),
        "vmax": float(np.sqrt((v ** 2).sum(1)).max()),
        "rho_mean": float(P[:, 9].mean()),
        "ymax": float(P[:, 1].max()),
    }


RHO_BINS = np.linspace(1.0, 2.0, 11)


def rho_histogram(P):
    """Share of the particles per bin of rho / rho0 (10 bins over [1, 2]; the density floor puts free particles at 1)."""
    return np.histogram(np.clip(P[:, 9] / P[:, 11], 1.0, 2.0 - 1e-9), bins=RHO_BINS)[0] / len(P)


def test_config1_1000_frames_bounded_drift():
    """Bounded drift, quantitatively: next to the engine and the oracle a THIRD run -- the oracle again, started with
    every position one ulp off -- measures how far two runs of this chaotic system drift apart by themselves.  Every
    50 frames the engine may differ from the oracle, in the vertical centre of mass, in the density histogram (10 bins
    of rho / rho0, L1 distance) by at most 3 x the largest distance the two ORACLE runs have shown up to that frame (+ a
    small floor for the first frames, where the twins still coincide), and in the kinetic energy by at most 8 x (one
    twin is one sample of the chaos, and the kinetic energy of the sloshing tank is its noisiest aggregate: 4.4 %
    engine-oracle against 0.7 % oracle-twin at frame 350, 4.6 % against 5.5 % at frame 400).  Measured on
    MI355X (scripts/freerun_probe2.py): engine-oracle and oracle-oracle spreads are the same size throughout
    (com_y <= 1.5e-3 vs 1.6e-3 units, histogram L1 <= 0.13 vs 0.15 from frame 100 on)."""
    from oracle.ora import Oracle
    eng, ora, twin = abi.Engine(0), Oracle(omp=True), Oracle(omp=True)
    se, so, st = scenes.config1(eng), scenes.config1(ora), scenes.config1(twin)
    se.sph.use_step_frame = True
    se.start(); so.start(); st.start()
    P0 = twin.download_particles()
    P0[:, 0:3] = np.nextafter(P0[:, 0:3], np.float32(np.inf))
    twin.upload_particles(P0)
    checkpoints = {}
    env = dict(comy=0.0, hist=0.0, ke=0.0)                 # running maxima of the oracle-vs-twin distances
    floor = dict(comy=2e-4, hist=5e-3, ke=1e-2)
    factor = dict(comy=3.0, hist=3.0, ke=8.0)
    trajectory = []
    for f in range(1, 1001):
        se.frame(); so.frame(); st.frame()
        if f in (10, 100, 300, 1000):
            checkpoints[f] = (aggregates(eng.download_particles()), aggregates(ora.download_particles()))
        if f % 50 == 0:
            P, Q, T = eng.download_particles(), ora.download_particles(), twin.download_particles()
            a, b, c = aggregates(P), aggregates(Q), aggregates(T)
            dist = lambda x, X, y, Y: dict(comy=abs(x["com"][1] - y["com"][1]), hist=float(np.abs(rho_histogram(X) - rho_histogram(Y)).sum()),
                                           ke=abs(x["ke"] - y["ke"]) / max(y["ke"], 1e-9))          # noqa: E731
            chaos, got = dist(c, T, b, Q), dist(a, P, b, Q)
            for k in env:
                env[k] = max(env[k], chaos[k])
                assert got[k] <= factor[k] * env[k] + floor[k], \
                    f"frame {f}: engine-oracle {k} distance {got[k]:.4g} > {factor[k]} x the oracle's own one-ulp spread {env[k]:.4g} + {floor[k]}"
            trajectory.append((f, round(a["com"][1], 4), round(b["com"][1], 4), round(got["hist"], 4), round(chaos["hist"], 4)))
    box = 10.0
    for f, (a, b) in checkpoints.items():
        assert np.all(np.isfinite(list(a["com"]) + [a["ke"], a["vmax"]])), f"frame {f}: engine state not finite"
        # bounds: early frames are still deterministic-ish, late ones have decorrelated particle-wise
        tight = f <= 10
        assert np.abs(a["com"] - b["com"]).max() <= (1e-4 if tight else 0.02) * box, (f, a["com"], b["com"])
        assert abs(a["pe"] - b["pe"]) <= (1e-4 if tight else 0.03) * abs(b["pe"]), (f, a["pe"], b["pe"])
        assert abs(a["rho_mean"] - b["rho_mean"]) <= (1e-4 if tight else 0.03) * b["rho_mean"], (f, a["rho_mean"], b["rho_mean"])
        if tight:
            assert abs(a["ke"] - b["ke"]) <= 1e-3 * b["ke"], (f, a["ke"], b["ke"])
        assert abs(a["ymax"] - b["ymax"]) <= (1e-3 if tight else 0.15) * box, (f, a["ymax"], b["ymax"])
    # the fluid has settled into the lower part of the box on both sides
    a, b = checkpoints[1000]
    assert a["com"][1] < 3.0 and b["com"][1] < 3.0
    print("frame, com_y engine, com_y oracle, histogram L1 engine-oracle, oracle-twin:", trajectory)


def test_scene4_two_phase_obb_obstacle_600_frames_bounded_drift_per_phase():
    """The same regime for the reference's Scene 4 (two phases, OBB boundary cube, orbiting sphere obstacle) at 5 % of its
    particle count: engine and oracle run 600 frames from the same spawn state without re-synchronisation; aggregates are
    compared PER PHASE.  Bounds are ~3 x the deviations observed on MI355X (scripts/freerun_probe.py): identical to 1e-4
    at frame 10, 1e-3-ish at frame 100, chaotic decorrelation afterwards (the tank is 20 x 30 x 20 units)."""
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle(omp=True)
    builder = lambda e: scenes.unity_scene(e, which=4, scale_count=0.05)      # noqa: E731
    se, so = builder(eng), builder(ora)
    se.sph.use_step_frame = True
    se.start(); so.start()
    box = 20.0
    bounds = {10: dict(com=1e-5, ke=1e-4, pe=1e-4, rho=1e-4), 100: dict(com=1e-3, ke=0.01, pe=0.01, rho=0.01),
              300: dict(com=0.02, ke=0.25, pe=0.03, rho=0.03), 600: dict(com=0.03, ke=0.30, pe=0.03, rho=0.03)}
    pe0 = {}
    for f in range(1, 601):
        se.frame(); so.frame()
        if f not in bounds:
            continue
        P, Q = eng.download_particles(), ora.download_particles()
        assert np.isfinite(P).all() and eng.read_counter(abi.CTR_NAN) == 0
        assert np.array_equal(P[:, 11:14], Q[:, 11:14])                      # static fields never drift
        for rho0 in (20.0, 10.0):
            a, b = aggregates(P[P[:, 11] == rho0]), aggregates(Q[Q[:, 11] == rho0])
            pe0.setdefault(rho0, abs(b["pe"]))                               # potential energy scale: the spawn height
            t = bounds[f]
            what = (f, rho0)
            assert np.abs(a["com"] - b["com"]).max() <= t["com"] * box, (what, a["com"], b["com"])
            assert abs(a["ke"] - b["ke"]) <= t["ke"] * b["ke"], (what, a["ke"], b["ke"])
            assert abs(a["pe"] - b["pe"]) <= t["pe"] * pe0[rho0], (what, a["pe"], b["pe"])
            assert abs(a["rho_mean"] - b["rho_mean"]) <= t["rho"] * b["rho_mean"], (what, a["rho_mean"], b["rho_mean"])
    eng.destroy()
