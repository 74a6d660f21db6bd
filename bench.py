#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the SPH frame step on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W

A "step" is one frame of SPH.Update (SPH.cs:201-222: grid build + subSteps x {half-step, clear,
density, force, full-step, boundary} + damping) over the resident particle set.  At N=1 the
workload is BASELINE config 3 (configs[2]): 1,048,576 particles, two-phase, orbiting sphere
obstacle, rocking boundary cube -- the 1M-particle configuration the metric is quoted on.
Particles are synthetic (seeded uniform spawn, ORACLE rule O9) and already resident in HBM
when the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FORCE_BYTES_PER_PARTICLE = 64   # R pos12 vel12 rho4 p4 visc4 mass4; W acc12 vel12
DENSITY_BYTES_PER_PARTICLE = 28


def measured_traffic(kernel, workload, n):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r1_traffic.json,
    collected with scripts/gpu_pmc.sh in separate --pmc runs); None if not measured for this workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        if t["workload"] != workload or t["particles"] != n:
            return None
        k = t["kernels"][kernel]
        return (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0
    except Exception:
        return None


def valu_issue(kernel, workload, n, launch_ms):
    """Secondary roofline of a neighbour kernel: VALU wave-instructions per launch (SQ_INSTS_VALU from the
    committed PMC pass of this workload) / live launch time, against the issue rate measured on this
    chip (scripts/experiments/valu_rate.hip).  These kernels are bound by candidate tests, not by HBM."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        if t["workload"] != workload or t["particles"] != n:
            return None
        insts = t["kernels"][kernel]["SQ_INSTS_VALU"]
        peak = t["valu_issue_peak"]["wave_insts_per_s_per_simd"] * t["valu_issue_peak"]["simds"] / 1e9
        ach = insts / (launch_ms * 1e-3) / 1e9
        return {"wave_insts_per_launch": insts, "achieved": ach, "peak": peak, "unit": "G wave-inst/s", "frac": ach / peak}
    except Exception:
        return None


def cpu_baseline(scene_builder, start_state, frames_done, budget_s=15.0):
    """Times the OpenMP oracle (kind "port") on the same workload, continuing from the state the
    GPU reached after warm-up, for a bounded number of frames."""
    from oracle.ora import Oracle
    ora = Oracle(omp=True)
    sc = scene_builder(ora)
    sc.start()
    for _ in range(frames_done):          # bring movers / time to the same frame
        for m in sc.movers:
            m.Update()
        sc.time.advance()
    ora.upload_particles(start_state)
    n = ora.n
    t0 = time.perf_counter()
    sc.frame()                            # untimed warm-up frame (page-faults the bins)
    t1 = time.perf_counter() - t0
    frames = int(max(1, min(30, budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(frames):
        sc.frame()
    dt = time.perf_counter() - t0
    ups = n * sc.sph.subSteps * frames / dt
    out = {"value": ups, "unit": "particle-updates/s", "cores": ora.threads(), "kind": "port",
           "sample": f"{frames} frames of the same {n}-particle workload from the post-warm-up state, "
                     f"OpenMP oracle (g++ -O2 -fopenmp, no -march), {dt:.1f} s"}
    ora.destroy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=0, help="override the particle count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="", help="1 GPU: config2|config3|config4|config5|config5narrow (default config3); "
                                                    "N GPUs: config5 (weak scaling, default) | config4 (strong scaling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from sph_amd import abi, scenes

    if args.gpus > 1:
        from sph_amd import slab
        return slab.bench_main(args, rank, local_rank, world)

    wl = args.workload or "config3"
    builders = {
        "config2": lambda e: scenes.config2(e, n=args.particles or 262144),
        "config3": lambda e: scenes.config3(e, n=args.particles or 1048576),
        "config4": lambda e: scenes.config4(e, n=args.particles or 16777216),
        "config5": lambda e: scenes.config5(e, gpus=1, per_gpu=args.particles or 8388608),
        "config5narrow": lambda e: scenes.config5(e, gpus=1, per_gpu=args.particles or 8388608, block=(20.0, 256.0)),
    }
    builder = builders[wl]
    eng = abi.Engine(local_rank)
    sc = builder(eng)
    sc.sph.use_step_frame = True
    sc.start()
    n = eng.n
    for _ in range(args.warmup):
        sc.frame()
    eng.synchronize()
    try:
        eng.get_timings()                 # drop warm-up frames from the stage averages
    except abi.SphError:
        pass
    start_state = None if args.no_cpu_baseline else eng.download_particles()

    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc.frame()
    eng.synchronize()
    dt = time.perf_counter() - t0
    tm = eng.get_timings()                # hipEvents on the engine's stream, averaged over the K frames
    overflow = eng.read_counter(abi.CTR_BIN_OVERFLOW)

    sub = sc.sph.subSteps
    value = n * sub * args.steps / dt
    force_s = tm["force_ms"] * 1e-3 / 1.0
    achieved = FORCE_BYTES_PER_PARTICLE * n / force_s / 1e9
    dens_gbps = DENSITY_BYTES_PER_PARTICLE * n / (tm["density_ms"] * 1e-3) / 1e9
    result = {
        "metric": "particle-updates/sec",
        "value": value,
        "unit": "particle-updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{wl}: {n} particles, two-phase, subSteps {sub}, "
                               f"grid {sc.sph.gridResolutionX}x{sc.sph.gridResolutionY}x{sc.sph.gridResolutionZ}",
                   "particles": n, "sub_steps": sub, "parallelism": "1 GPU",
                   "bin_overflow_cells": overflow},
        "roofline": {"bound": "hbm", "kernel": "k_force_xsph_hits (CS_ForceXSPH)", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": measured_traffic("k_force_xsph_hits", wl, n),
                     "alg_bytes_per_particle": FORCE_BYTES_PER_PARTICLE, "avg_launch_ms": tm["force_ms"],
                     "valu_issue": valu_issue("k_force_xsph_hits", wl, n, tm["force_ms"]),
                     "density_kernel": {"achieved": dens_gbps, "frac": dens_gbps / HBM_PEAK_GBPS,
                                        "alg_bytes_per_particle": DENSITY_BYTES_PER_PARTICLE,
                                        "avg_launch_ms": tm["density_ms"],
                                        "traffic": measured_traffic("k_density_pressure_rows", wl, n),
                                        "valu_issue": valu_issue("k_density_pressure_rows", wl, n, tm["density_ms"])}},
        "stage_ms": {k: round(v, 4) for k, v in tm.items() if k.endswith("_ms")},
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(builder, start_state, args.warmup)
    eng.destroy()
    print(json.dumps(result))


if __name__ == "__main__":
    main()
