#!/usr/bin/env python
"""Condense a rocprofv3 kernel_stats CSV (or a counter_collection CSV) into the small summaries kept under profiles/.
Usage: summarize_pmc.py <csv> <out.csv> [top N]"""
import collections
import csv
import re
import sys


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)                 # drop the argument list
    if n.startswith("at::native::"):
        n = "aten:" + re.sub(r"<.*", "", n[len("at::native::"):])
    if n.startswith("Cijk_"):
        n = "hipBLASLt:" + n[:40]
    return n[:90]


def main(path, out, top=60):
    rows = list(csv.DictReader(open(path)))
    if rows and "Counter_Name" in rows[0]:
        agg = collections.OrderedDict()
        for r in rows:
            agg.setdefault((short(r["Kernel_Name"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        with open(out, "w") as f:
            f.write("kernel,counter,mean_per_dispatch,dispatches\n")
            for (k, c), v in agg.items():
                f.write(f'"{k}",{c},{sum(v)/len(v):.3f},{len(v)}\n')
        return
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0])
        a[0] += int(r["Calls"])
        a[1] += int(r["TotalDurationNs"])
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct_of_all_gpu_time\n")
        for k, (calls, ns) in items[:top]:
            f.write(f'"{k}",{calls},{ns/1e6:.3f},{ns/1e3/calls:.2f},{100.0*ns/tot:.2f}\n')
        rest = items[top:]
        if rest:
            ns = sum(v[1] for _, v in rest)
            f.write(f'"({len(rest)} more kernels)",{sum(v[0] for _, v in rest)},{ns/1e6:.3f},,{100.0*ns/tot:.2f}\n')


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
