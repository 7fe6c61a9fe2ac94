"""Read a rocprofv3 kernel_trace.csv of tools/refloop_profile.py --trace-only: the last loop's forward() calls are
delimited by their project_rays_kernel launch.  Prints wall / busy time of the loop, per-kernel sums over its calls and
the launch-by-launch timeline (duration, gap to the previous kernel's end) of one mid-loop call.
Usage: refloop_trace.py <kernel_trace.csv> [calls per loop = 18]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 18
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:100]


marks = [i for i, r in enumerate(rows) if "project_rays_kernel" in r[2]]
lo = marks[-ncall]
loop = rows[lo:]
wall = (loop[-1][1] - loop[0][0]) / 1e6
busy = sum(e - s for s, e, _ in loop) / 1e6
print(f"last loop: {len(loop)} launches, wall {wall:.2f} ms, kernel busy {busy:.2f} ms, idle {wall - busy:.2f} ms")
agg, cnt = defaultdict(float), defaultdict(int)
for s, e, n in loop:
    agg[short(n)] += (e - s) / 1e6
    cnt[short(n)] += 1
for n, t in sorted(agg.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{t:9.3f} ms {100 * t / busy:5.1f}% x{cnt[n]:5d}  {n}")
mid = marks[-ncall // 2 - 1], marks[-ncall // 2]
print(f"--- one call ({mid[1] - mid[0]} launches, {(rows[mid[1]][0] - rows[mid[0]][0]) / 1e3:.1f} us start to next start)")
prev = rows[mid[0] - 1][1]
for s, e, n in rows[mid[0]:mid[1]]:
    print(f"{(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f} us  {short(n)}")
    prev = e
