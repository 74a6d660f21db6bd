# How many vector-L1 (TCP) accesses does a wave-wide 16-byte-per-lane load make, and how many can a CU make per clock?
# scripts/experiments/gather_rate (built beforehand on the CPU) under a counter pass; one row per launch pattern.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tcp_rate
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT -o pmc -- ./scripts/experiments/gather_rate > $OUT/run.txt 2> $OUT/run.err
python3 - <<'PY' | tee gpurun_out/tcp_rate/summary.txt
import csv
from collections import defaultdict
rows = list(csv.DictReader(open("gpurun_out/tcp_rate/pmc_counter_collection.csv")))
by = defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
names = ["contiguous", "same address", "runs of 11 / 132 B", "runs of 11 / 176 B", "stride 64 B", "random", "runs of 11 random"]
loads = 256 * 8 * 4 * 512 * 8          # wave-loads per launch
print(f"{'pattern':22s} {'TCP accesses / wave-load':>26s} {'kernel clk / wave-load / CU':>28s} {'TCP accesses / clk / CU':>24s} {'TA busy':>8s}")
for i, d in enumerate(sorted(by)):
    v = by[d]
    if i % 3 != 2:
        continue
    clk = v["GRBM_GUI_ACTIVE"] / 8.0
    acc = v["TCP_TOTAL_CACHE_ACCESSES_sum"]
    print(f"{names[i // 3]:22s} {acc / loads:26.2f} {clk / (loads / 256):28.2f} {acc / 256 / clk:24.3f} {v['TA_TA_BUSY_sum'] / 256 / clk:8.2f}")
PY
cat $OUT/run.txt
