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


def test_config1_1000_frames_bounded_drift():
    from oracle.ora import Oracle
    eng, ora = abi.Engine(0), Oracle(omp=True)
    se, so = scenes.config1(eng), scenes.config1(ora)
    se.sph.use_step_frame = True
    se.start(); so.start()
    checkpoints = {}
    for f in range(1, 1001):
        se.frame(); so.frame()
        if f in (10, 100, 300, 1000):
            checkpoints[f] = (aggregates(eng.download_particles()), aggregates(ora.download_particles()))
    box = 10.0
    for f, (a, b) in checkpoints.items():
        assert np.all(np.isfinite(list(a["com"]) + [a["ke"], a["vmax"]])), f"frame {f}: engine state not finite"
        # bounds: early frames are still deterministic-ish, late ones have decorrelated particle-wise
        tight = f <= 10
        assert np.abs(a["com"] - b["com"]).max() <= (1e-4 if tight else 0.02) * box, (f, a["com"], b["com"])
        assert abs(a["pe"] - b["pe"]) <= (1e-4 if tight else 0.03) * abs(b["pe"]), (f, a["pe"], b["pe"])
        assert abs(a["rho_mean"] - b["rho_mean"]) <= (1e-4 if tight else 0.03) * b["rho_mean"], (f, a["rho_mean"], b["rho_mean"])
        # kinetic energy: 35 % relative + a noise floor of 0.25 m/s rms (the residual jitter of a
        # settled tank is chaotic: 0.23 vs 0.34 m/s rms at frame 1000 is the same physical state)
        ke_floor = 0.0 if tight else 0.5 * 16384 * 0.2 * 0.25 ** 2
        assert abs(a["ke"] - b["ke"]) <= (1e-3 if tight else 0.35) * b["ke"] + ke_floor, (f, a["ke"], b["ke"])
        assert abs(a["ymax"] - b["ymax"]) <= (1e-3 if tight else 0.15) * box, (f, a["ymax"], b["ymax"])
    # the fluid has settled into the lower part of the box on both sides
    a, b = checkpoints[1000]
    assert a["com"][1] < 3.0 and b["com"][1] < 3.0
    print({f: {k: (np.round(x[k], 4), np.round(y[k], 4)) for k in ("ke", "pe", "vmax", "rho_mean", "ymax")}
           for f, (x, y) in checkpoints.items()})
