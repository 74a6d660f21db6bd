"""Per-kernel duration over the TIMED region of a profiled bench run.

`rocprofv3 --stats` averages every dispatch of the run, warm-up frames included, and the warm-up
frames of the bench workload are much heavier (particles start at spawn density).  bench.py's
hipEvent averages cover the timed frames only, so this script reduces the kernel trace of the same
run to the last K dispatches of each kernel -- the figure to compare with bench.py's `stage_ms`.

    python3 scripts/trace_tail_stats.py <trace_kernel_trace.csv> <K>
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel,calls_total,last_k,mean_us_last_k,min_us_last_k,max_us_last_k,mean_us_all")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1][-K:])):
    t = v[-K:]
    print(f"{k},{len(v)},{len(t)},{sum(t)/len(t)/1e3:.2f},{min(t)/1e3:.2f},{max(t)/1e3:.2f},{sum(v)/len(v)/1e3:.2f}")
