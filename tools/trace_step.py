"""Aggregate a rocprofv3 kernel_trace.csv over ONE steady-state step: the interval between the last two launches of a
marker kernel (one launch per step).  Usage: trace_step.py <kernel_trace.csv> <marker substring> [top N] [--launches]
(--launches: the longest individual launches of that step, in time order of their rank, instead of per-kernel sums)"""
import csv
import re
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
path, marker = args[0], args[1]
top = int(args[2]) if len(args) > 2 else 40
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = marks[-2], marks[-1]
if "--launches" in sys.argv:
    t0 = rows[lo][0]
    for s, e, n in sorted(rows[lo:hi], key=lambda r: r[0] - r[1])[:top]:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        print(f"{(e - s) / 1e6:8.3f} ms at {(s - t0) / 1e6:8.2f} ms  {re.sub(r'^void ', '', n)[:110]}")
    sys.exit(0)
agg, cnt = defaultdict(float), defaultdict(int)
for s, e, n in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)[:100]
    agg[n] += (e - s) / 1e6
    cnt[n] += 1
wall = (rows[hi][0] - rows[lo][0]) / 1e6
busy = sum(agg.values())
print(f"step wall {wall:.2f} ms, kernel busy {busy:.2f} ms, {hi - lo} launches")
for n, t in sorted(agg.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{t:9.3f} ms {100 * t / busy:5.1f}% x{cnt[n]:5d}  {n}")
