"""profiles/traffic.json from the counter summaries under profiles/r<N>_pmc/ (scripts/gpu_r3_pmc.sh -> gpu_pmc.sh ->
pmc_summary.py): one run per (workload, particles, warm-up), which is how bench.py looks them up.
    python3 scripts/make_traffic_json.py [round=3]      (keeps the runs of the earlier rounds, newest first)"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
ROUND = int(sys.argv[1]) if len(sys.argv) > 1 else 3
PMC = os.path.join(ROOT, "profiles", f"r{ROUND}_pmc")
RUNS = [("w5", "config3", 1048576, 5, 20), ("w20", "config3", 1048576, 20, 30), ("c5w5", "config5", 8388608, 5, 20)]
RENAME = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB"}


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = re.sub(r"^void ", "", line.strip())
            cur = re.sub(r"<.*>$", "", cur).replace("sph::", "")
            out.setdefault(cur, {})
        else:
            k, v = line.split()[:2]
            out[cur][RENAME.get(k, k)] = float(v)
    return out


old = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
runs = [r for r in old["runs"] if r.get("round") != ROUND]
for tag, wl, n, warm, steps in RUNS:
    kernels = {}
    for part in ("sq", "fetch", "write", "ta"):
        for k, d in parse(os.path.join(PMC, f"{tag}_{part}_summary.txt")).items():
            kernels.setdefault(k, {}).update(d)
    runs.append({"workload": wl, "particles": n, "warmup": warm, "steps": steps, "round": ROUND,
                 "source": f"profiles/r{ROUND}_pmc/{tag}_{{sq,fetch,write,ta}}_summary.txt: python3 bench.py --workload {wl} --steps {steps} "
                           f"--warmup {warm}, mean over the {steps} timed frames",
                 "kernels": kernels})
# bench.py takes the first match: newest round first
runs.sort(key=lambda r: -r["round"])
old["runs"] = runs
json.dump(old, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("runs:", [(r["workload"], r["particles"], r["warmup"], r["round"]) for r in runs])
