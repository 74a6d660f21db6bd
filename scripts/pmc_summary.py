"""Per-kernel mean of each counter from a rocprofv3 --pmc CSV over the last N dispatches of each kernel
(N = argv[2], default 5: the timed frames of the bench run)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
by = defaultdict(lambda: defaultdict(list))
for r in rows:
    by[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(by.items()):
    if not any(x in k for x in ("force", "density", "permute", "hash_count", "place", "rank", "cum_", "full_bound")):
        continue
    print(k)
    for c, v in sorted(d.items()):
        tail = v[-N:]
        print(f"    {c:28s} {sum(tail)/len(tail):16.1f}  (n={len(v)})")
