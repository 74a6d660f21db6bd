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
