"""Measurement helpers shared by bench.py (one GPU) and sph_amd.slab.bench_main (N GPUs): evidence from the
committed counter passes (profiles/traffic.json), keyed by the run they were collected on.  No GPU calls."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FORCE_BYTES_PER_PARTICLE = 64   # R pos12 vel12 rho4 p4 visc4 mass4; W acc12 vel12
DENSITY_BYTES_PER_PARTICLE = 28
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")
# what the counters say limits the two neighbour kernels; `bound` stays the contract's
# roofline (the one `peak` and `frac` are priced against)
MEASURED_LIMITER = ("vector-memory path (texture addresser + vector-L1 accesses of the candidate gathers) + VALU issue, not HBM: "
                    "see valu_issue, ta_busy_frac, l1_accesses_per_clk_per_cu")


# ------------------------------------------------------------------------------------------------
def counter_run(workload, n, warmup, path=TRAFFIC_FILE):
    """The committed PMC run (profiles/traffic.json, scripts/gpu_pmc.sh) taken on THIS workload, particle count
    and warm-up; None if there is none (then the fields that need counters are null: a count collected in another
    state of the tank -- e.g. 20 warm-up frames vs the blow-up at 5 -- is not evidence for this run)."""
    try:
        for run in json.load(open(path))["runs"]:
            if run["workload"] == workload and run["particles"] == n and run["warmup"] == warmup:
                return run
    except Exception:
        pass
    return None


def measured_traffic(run, kernel):
    """HBM bytes per launch: (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB (gfx950 correction, MI355X_MICROARCH.md)."""
    try:
        k = run["kernels"][kernel]
        return (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0
    except Exception:
        return None


def valu_issue(run, kernel, launch_ms, path=TRAFFIC_FILE):
    """Secondary roofline of a neighbour kernel: VALU wave-instructions per launch (SQ_INSTS_VALU of the matching
    counter run) / live launch time, against the issue rate measured on this chip (scripts/experiments/valu_rate.hip)."""
    try:
        pk = json.load(open(path))["valu_issue_peak"]
        insts = run["kernels"][kernel]["SQ_INSTS_VALU"]
        if not launch_ms:
            return None
        peak = pk["wave_insts_per_s_per_simd"] * pk["simds"] / 1e9
        ach = insts / (launch_ms * 1e-3) / 1e9
        return {"wave_insts_per_launch": insts, "achieved": ach, "peak": peak, "unit": "G wave-inst/s", "frac": ach / peak,
                "counters_from": f"warmup {run['warmup']}, steps {run['steps']} ({run.get('source', 'profiles/')})"}
    except Exception:
        return None


def ta_busy_frac(run, kernel, launch_ms=None, cus=256):
    """Share of the kernel's time the texture addressers were busy, from ONE counter run: TA_TA_BUSY_sum (busy cycles
    summed over the CUs) / CUs, over the kernel's own cycles GRBM_GUI_ACTIVE / 8 of that same run (rocprofv3 sums the
    counter over the 8 XCDs).  No assumed clock, no launch time of another run.  Runs collected before round 3 carry no
    GRBM_GUI_ACTIVE: for those the old estimate (this run's launch time at a nominal 2.4 GHz) is returned."""
    try:
        k = run["kernels"][kernel]
        busy = k["TA_TA_BUSY_sum"] / cus
        if "GRBM_GUI_ACTIVE" in k:
            return busy / (k["GRBM_GUI_ACTIVE"] / 8.0)
        return busy / (2.4e9 * launch_ms * 1e-3)
    except Exception:
        return None


def l1_access_rate(run, kernel, cus=256):
    """Vector-L1 (TCP) accesses per clock and CU over the kernel's own cycles, from ONE counter run
    (TCP_TOTAL_CACHE_ACCESSES_sum / CUs over GRBM_GUI_ACTIVE / 8).  The path sustains 1.0-1.3 per clock on synthetic
    gathers with the addresser 98 % busy (profiles/r3_wave_trace/tcp_access_rate_microbench.txt)."""
    try:
        k = run["kernels"][kernel]
        return k["TCP_TOTAL_CACHE_ACCESSES_sum"] / cus / (k["GRBM_GUI_ACTIVE"] / 8.0)
    except Exception:
        return None
