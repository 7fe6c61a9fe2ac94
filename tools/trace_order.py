"""Time-ordered launch list of ONE steady-state step from a rocprofv3 kernel_trace.csv (the interval between the last two
launches of a marker kernel): index, start offset, duration, gap to the previous kernel's end, short name.
Usage: trace_order.py <kernel_trace.csv> <marker substring>"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r[2]]
lo, hi = marks[-2], marks[-1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "aten:", n)
    n = re.sub(r"\(.*", "", n)
    return n[:70]


t0, prev = rows[lo][0], rows[lo][0]
for i, (s, e, n) in enumerate(rows[lo:hi]):
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:7.1f}  {short(n)}")
    prev = e
