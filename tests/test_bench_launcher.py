"""`python bench.py --gpus N` started plainly (WORLD_SIZE unset) must start its own N ranks: the plumbing, on CPU.
The ranks themselves need GPUs; here they are replaced by small child programs that report the environment
they were given."""
import json
import os
import sys

import pytest

import bench
from sph_amd import benchutil

CHILD = ("import json, os, sys; r = int(os.environ['RANK']); "
         "print(json.dumps({'metric': 'm', 'rank': r, 'local': os.environ['LOCAL_RANK'], 'world': os.environ['WORLD_SIZE'], "
         "'addr': os.environ['MASTER_ADDR'], 'port': os.environ['MASTER_PORT'], 'argv': sys.argv[1:]})); "
         "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r and 3 or 0)")


def _child_plan(n, argv=(), extra_env=None):
    plan = bench.launch_plan(list(argv), n, port=29400)
    out = []
    for cmd, env in plan:
        env = dict(env)
        env.update(extra_env or {})
        out.append(([cmd[0], "-c", CHILD] + cmd[3:], env))     # same interpreter, arguments and environment; stub program
    return out


def test_launch_plan_command_and_environment():
    plan = bench.launch_plan(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, port=29512, base_env={"PATH": "/bin"})
    assert len(plan) == 4
    for r, (cmd, env) in enumerate(plan):
        assert cmd[0] == sys.executable and cmd[2] == os.path.abspath(bench.__file__)
        assert cmd[3:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == "4" and env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29512"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # dmabuf IPC for RCCL
        assert env["PATH"] == "/bin"
    assert len({id(env) for _, env in plan}) == 4                # one dict per rank


def test_free_port_is_usable():
    import socket
    p = bench.free_port()
    with socket.socket() as s:
        s.bind(("127.0.0.1", p))


def test_run_ranks_relays_rank0_line_only():
    rc, out = bench.run_ranks(_child_plan(3, ["--gpus", "3"]))
    assert rc == 0
    line = bench.pick_json_line(out)
    d = json.loads(line)
    assert d["rank"] == 0 and d["world"] == "3" and d["addr"] == "127.0.0.1" and d["argv"] == ["--gpus", "3"]
    assert out.count("\n") == 1                                   # the other ranks' stdout went to stderr


def test_failing_rank_fails_the_launch():
    rc, _ = bench.run_ranks(_child_plan(2, extra_env={"FAIL_RANK": "1"}))
    assert rc == 3


def test_main_self_launches_only_without_world_size(monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "run_ranks", lambda plan: (calls.append(plan), (0, '{"metric": "x", "n_gpus": 2}\n'))[1])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "2", "--steps", "1", "--warmup", "0"]) == 0
    assert len(calls) == 1 and len(calls[0]) == 2
    # a launch that yields no result line is an error, not a silent success
    monkeypatch.setattr(bench, "run_ranks", lambda plan: (0, "noise\n"))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert e.value.code != 0
    # under a launcher the process is a rank: world size and --gpus must agree
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "2"])


def test_pick_json_line_skips_noise():
    text = 'warming up\n{"not": "it"}\n{"metric": "particle-updates/sec", "value": 1}\ntrailing\n'
    assert json.loads(bench.pick_json_line(text))["value"] == 1
    assert bench.pick_json_line("nothing here") is None


def test_counter_runs_are_keyed_by_warmup(tmp_path):
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps({"valu_issue_peak": {"wave_insts_per_s_per_simd": 1e9, "simds": 1000},
                             "runs": [{"workload": "config3", "particles": 100, "warmup": 20, "steps": 5,
                                       "kernels": {"k": {"FETCH_SIZE_KiB": 10.0, "WRITE_SIZE_KiB": 4.0, "SQ_INSTS_VALU": 2e6,
                                                         "TA_TA_BUSY_sum": 256 * 2.4e9 * 1e-4}}}]}))
    run = benchutil.counter_run("config3", 100, 20, path=str(f))
    assert benchutil.measured_traffic(run, "k") == (2 * 10.0 + 4.0) * 1024
    v = benchutil.valu_issue(run, "k", 1.0, path=str(f))
    assert abs(v["frac"] - 2e6 / 1e-3 / 1e12) < 1e-12 and "warmup 20" in v["counters_from"]
    assert abs(benchutil.ta_busy_frac(run, "k", 1.0) - 0.1) < 1e-9
    # counters collected at another warm-up (another state of the tank) are not evidence for this run
    assert benchutil.counter_run("config3", 100, 5, path=str(f)) is None
    assert benchutil.measured_traffic(None, "k") is None and benchutil.valu_issue(None, "k", 1.0, path=str(f)) is None


def test_counter_rates_use_the_counter_runs_own_clock():
    """Round 3: the texture-addresser share and the vector-L1 access rate divide by the kernel's own cycles of the SAME
    counter run (GRBM_GUI_ACTIVE, summed over 8 XCDs), not by another run's launch time at an assumed clock."""
    run = {"kernels": {"k": {"TA_TA_BUSY_sum": 256 * 750.0, "GRBM_GUI_ACTIVE": 8 * 1000.0,
                             "TCP_TOTAL_CACHE_ACCESSES_sum": 256 * 900.0}}}
    assert abs(benchutil.ta_busy_frac(run, "k", launch_ms=123.0) - 0.75) < 1e-12       # the launch time is not used
    assert abs(benchutil.l1_access_rate(run, "k") - 0.9) < 1e-12
    assert benchutil.l1_access_rate(run, "other") is None and benchutil.l1_access_rate(None, "k") is None
    # the committed round-3 runs carry both counters for the two neighbour kernels
    t = json.load(open(benchutil.TRAFFIC_FILE))
    r3 = [r for r in t["runs"] if r.get("round") == 3 and r["workload"] == "config3"]
    assert r3
    for r in r3:
        for k in ("k_density_pressure_rows", "k_force_xsph_hits"):
            assert 0.3 < benchutil.l1_access_rate(r, k) < 1.5 and 0.3 < benchutil.ta_busy_frac(r, k) < 1.0


def test_committed_traffic_file_is_well_formed():
    t = json.load(open(benchutil.TRAFFIC_FILE))
    assert t["runs"], "profiles/traffic.json has no run"
    for run in t["runs"]:
        assert {"workload", "particles", "warmup", "steps", "kernels"} <= set(run)
