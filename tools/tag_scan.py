"""Per-kernel sums of the L1 tag-conflict / TA-stall counters over one profiled command (which kernels read row-major memory in
an MFMA operand layout?).   python tools/tag_scan.py <dir with *counter_collection.csv> [top]"""
import collections
import csv
import glob
import os
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"].startswith("TCP_READ_TAGCONFLICT"):
            cnt[k] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0))[:top]
print(f"{'kernel':70s} {'calls':>6s} {'tagconflict Mclk':>17s} {'TA addr stalled Mclk':>21s} {'L1 accesses M':>14s} {'TA busy Mclk':>13s}")
for k, c in rows:
    print(f"{k:70s} {cnt[k]:6d} {c.get('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum', 0) / 1e6:17.1f} {c.get('TA_ADDR_STALLED_BY_TC_CYCLES_sum', 0) / 1e6:21.1f} "
          f"{c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / 1e6:14.1f} {c.get('TA_TA_BUSY_sum', 0) / 1e6:13.1f}")
