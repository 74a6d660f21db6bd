#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the SPH frame step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one frame of SPH.Update (SPH.cs:201-222: grid build + subSteps x {half-step, clear,
density, force, full-step, boundary} + damping) over the resident particle set.  At N=1 the
workload is BASELINE config 3 (configs[2]): 1,048,576 particles, two-phase, orbiting sphere
obstacle, rocking boundary cube -- the 1M-particle configuration the metric is quoted on.
Particles are synthetic (seeded uniform spawn, ORACLE rule O9) and already resident in HBM
when the timed region starts.  Prints ONE JSON line on rank 0.

N > 1: one process per GPU.  Under a launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) this process is one rank.  Started plainly (`python bench.py --gpus N`, WORLD_SIZE unset) it
is the PARENT: it starts N rank processes of itself with that environment -- before it has made any GPU,
HIP or torch call, and without ever replacing a process image -- relays rank 0's JSON line and exits
non-zero if any rank failed (launch_plan / run_ranks below; tests/test_bench_launcher.py).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)



# ------------------------------------------------------------------------------------------------
# self-launch of the N ranks
# ------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_plan(argv, n_ranks, port=None, python=None, script=None, base_env=None):
    """[(cmd, env)] for the n_ranks processes of one node: the same command line, torchrun's environment
    contract (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT)."""
    port = int(port or free_port())
    python = python or sys.executable
    script = script or os.path.abspath(__file__)
    base = dict(os.environ if base_env is None else base_env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    plan = []
    for r in range(n_ranks):
        env = dict(base)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPH_BENCH_SELF_LAUNCHED="1")
        plan.append(([python, "-u", script] + list(argv), env))
    return plan


def run_ranks(plan, poll_s=0.2):
    """Start every rank as a CHILD process, wait for all.  Rank 0's stdout is captured, the other ranks' goes to
    stderr.  If a rank fails, the others are ended (their exact PIDs) and its code is returned.  Returns
    (return_code, rank0_stdout)."""
    procs = []
    for r, (cmd, env) in enumerate(plan):
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(len(procs)))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with {code}; ending the other ranks", file=sys.stderr)
                for q in live:
                    procs[q].terminate()
        if live:
            time.sleep(poll_s)
    for p in procs:                 # a rank that ignored SIGTERM
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    return rc, (out0[0] if out0 else "")


def pick_json_line(text):
    """The last line of `text` that is a JSON object with a "metric" key (rank 0's result line)."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if "metric" in d:
                return line
    return None


def self_launch(argv, n_ranks):
    rc, out = run_ranks(launch_plan(argv, n_ranks))
    line = pick_json_line(out)
    if line is not None:
        print(line)                 # also of a run that declared itself invalid (value null, invalid_reason): the record says why
    if rc != 0 or line is None:
        sys.stderr.write(out)
        raise SystemExit(rc or 1)
    return 0


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
                     f"OpenMP oracle (g++ -O2 -fopenmp, no -march: built in the CPU container, runs on the GPU box's "
                     f"different host CPU), {dt:.1f} s"}
    ora.destroy()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=int, default=0, help="override the particle count (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="", help="1 GPU: config2|config3|config4|config5|config5narrow (default config3); "
                                                    "N GPUs: config5 (weak scaling, default) | config4 (strong scaling)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only starts the ranks (nothing GPU-related is imported yet)
        return self_launch(argv, args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from sph_amd import abi, scenes
    from sph_amd.benchutil import (DENSITY_BYTES_PER_PARTICLE, FORCE_BYTES_PER_PARTICLE, HBM_PEAK_GBPS, MEASURED_LIMITER,
                                   counter_run, l1_access_rate, measured_traffic, ta_busy_frac, valu_issue)

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
    # kernel times: start / stop events the library attaches to its own launches, on its own stream, on every timed
    # frame (they add no packet to the stream: sph_timings in include/sph_hip.h)
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
    try:
        tm = eng.get_timings()            # launch-attached events on the engine's stream, averaged over the timed frames
    except abi.SphError:                  # SPH_HIP_TIMING=0: no kernel times, the line still carries the throughput
        tm = {k: None for k in ("build_grid_ms", "half_step_ms", "density_ms", "force_ms", "full_step_ms",
                                "bound_obs_ms", "damping_ms", "frame_ms")}
        tm["frames"] = 0
    overflow = eng.read_counter(abi.CTR_BIN_OVERFLOW)
    non_finite = eng.read_counter(abi.CTR_NAN)

    sub = sc.sph.subSteps
    value = n * sub * args.steps / dt
    # without kernel times (SPH_HIP_TIMING=0) everything derived from them is null: the line stays strict JSON
    timed = bool(tm["frames"]) and tm["force_ms"] and tm["density_ms"]
    achieved = FORCE_BYTES_PER_PARTICLE * n / (tm["force_ms"] * 1e-3) / 1e9 if timed else None
    dens_gbps = DENSITY_BYTES_PER_PARTICLE * n / (tm["density_ms"] * 1e-3) / 1e9 if timed else None
    frac = (lambda g: None if g is None else g / HBM_PEAK_GBPS)
    # what `traffic` counts: FETCH_SIZE / WRITE_SIZE are the L2's fabric-side requests; Infinity-Cache hits are in them
    # (MI355X_MICROARCH.md, HBM section), and a state of ~150 B per particle fits the 256 MiB Infinity Cache up to ~1.5 M
    # particles -- there `traffic` is L2-miss traffic (an upper bound of HBM traffic), beyond it it is HBM traffic
    traffic_is = ("L2-miss (fabric-side) bytes per launch, Infinity-Cache hits included: the state fits the 256 MiB Infinity Cache, "
                  "so this is an upper bound of the HBM traffic" if n * 150 < 200e6 else
                  "L2-miss (fabric-side) bytes per launch = HBM traffic (the state does not fit the Infinity Cache)")
    run = counter_run(wl, n, args.warmup)
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
                   "bin_overflow_cells": overflow, "non_finite_particles": non_finite,
                   "frames_with_kernel_events": tm["frames"]},
        "roofline": {"bound": "hbm", "measured_limiter": MEASURED_LIMITER,
                     "kernel": "k_force_xsph_hits (CS_ForceXSPH)", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac(achieved),
                     "traffic": measured_traffic(run, "k_force_xsph_hits"), "traffic_is": traffic_is,
                     "traffic_from": None if run is None else f"warmup {run['warmup']}, steps {run['steps']} ({run.get('source', 'profiles/')})",
                     "alg_bytes_per_particle": FORCE_BYTES_PER_PARTICLE, "avg_launch_ms": tm["force_ms"],
                     "valu_issue": valu_issue(run, "k_force_xsph_hits", tm["force_ms"]),
                     "ta_busy_frac": ta_busy_frac(run, "k_force_xsph_hits", tm["force_ms"]),
                     "l1_accesses_per_clk_per_cu": l1_access_rate(run, "k_force_xsph_hits"),
                     "density_kernel": {"achieved": dens_gbps, "frac": frac(dens_gbps),
                                        "alg_bytes_per_particle": DENSITY_BYTES_PER_PARTICLE,
                                        "avg_launch_ms": tm["density_ms"],
                                        "traffic": measured_traffic(run, "k_density_pressure_rows"),
                                        "valu_issue": valu_issue(run, "k_density_pressure_rows", tm["density_ms"]),
                                        "ta_busy_frac": ta_busy_frac(run, "k_density_pressure_rows", tm["density_ms"]),
                                        "l1_accesses_per_clk_per_cu": l1_access_rate(run, "k_density_pressure_rows")}},
        "stage_ms": {k: (None if v is None else round(v, 4)) for k, v in tm.items() if k.endswith("_ms")},
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(builder, start_state, args.warmup)
    eng.destroy()
    print(json.dumps(result, allow_nan=False))
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
