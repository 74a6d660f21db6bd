"""The bench line itself, on the GPU at a small particle count: every field of the driver's contract is there, the
numbers hang together, and the line is strict JSON.  -m gpu."""
import json

import pytest

import bench

pytestmark = pytest.mark.gpu


def _run(capsys, argv):
    bench.main(argv)
    line = bench.pick_json_line(capsys.readouterr().out)
    assert line is not None, "bench.py printed no JSON line"
    json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c} in the bench line"))
    return json.loads(line)


def test_one_gpu_bench_line_keeps_the_contract(capsys):
    steps, warmup, n = 4, 3, 32768
    d = _run(capsys, ["--steps", str(steps), "--warmup", str(warmup), "--particles", str(n)])
    assert d["metric"] == "particle-updates/sec" and d["unit"] == "particle-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "config3" in d["config"]["workload"] and d["config"]["particles"] == n and "model" not in d["config"]
    # value = particles x subSteps x steps / wall time
    assert abs(d["value"] - n * d["config"]["sub_steps"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / the launch time measured live (events on the library's own launches)
    assert abs(r["achieved"] - r["alg_bytes_per_particle"] * n / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert 0 < r["avg_launch_ms"] < d["ms_per_step"]                       # the dominant kernel fits into the frame
    assert r["traffic"] is None and r["traffic_from"] is None              # no committed counter pass at this size: not borrowed
    assert d["config"]["frames_with_kernel_events"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and "frames" in c["sample"]
    assert d["value"] > c["value"]


def test_counters_are_quoted_only_for_the_run_they_were_taken_on(capsys):
    """At the driver's arguments (config 3, 1,048,576 particles, 5 warm-up frames) the line carries the committed
    counter pass of exactly that run; two frames are enough to see it."""
    d = _run(capsys, ["--steps", "2", "--warmup", "5", "--no-cpu-baseline"])
    r = d["roofline"]
    assert "cpu_baseline" not in d
    assert r["traffic"] and "warmup 5" in r["traffic_from"] and "r3_pmc" in r["traffic_from"]
    assert 0.3 < r["ta_busy_frac"] < 1.0 and 0.3 < r["l1_accesses_per_clk_per_cu"] < 1.5
    assert r["traffic"] > r["alg_bytes_per_particle"] * d["config"]["particles"]      # wasted re-reads are visible
    assert "Infinity" in r["traffic_is"]


def test_two_rank_bench_line_self_launched(tmp_path):
    """`python bench.py --gpus 2` the way the driver may call it without torchrun: the parent starts the ranks; on a box
    with one GPU they share it (said so in config.rehearsal).  Slab decomposition of config 5's geometry at a small
    particle count; the line keeps the contract and reports what makes it trustworthy."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # the same transport wherever the test runs: both ranks on GPU 0, host-staged messages (the RCCL path needs two GPUs
    # and is the driver's scaling run, not a unit test)
    env["SPH_BENCH_SHARE_GPU"], env["SPH_BENCH_BACKEND"] = "1", "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--particles", "262144", "--steps", "3",
                          "--warmup", "3"], cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = bench.pick_json_line(out.stdout)
    assert line is not None
    d = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    c = d["config"]
    assert d["n_gpus"] == 2 and c["world_size"] == 2 and d["scaling"] == "weak" and d["metric"] == "particle-updates/sec"
    assert c["particles"] == 2 * 262144 and c["self_launched"] is True and c["parallelism"] == "slab2"
    assert d["value"] is not None and d["invalid_reason"] is None
    assert abs(d["value"] - c["particles"] * c["sub_steps"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert c["every_id_owned_once"] is True and c["halo_violations"] == 0 and c["band_misses"] == 0
    assert c["transfer_rounds_per_exchange_rank0"] <= 2.0 and c["host_waits_per_frame_rank0"] >= 3.0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["avg_launch_ms"] > 0
    assert c["rehearsal"] and "share" in c["rehearsal"] and c["backend"] == "gloo"
