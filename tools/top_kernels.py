"""Print the top-N rows of a rocprofv3 kernel_stats.csv with shortened kernel names."""
import csv
import re
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {len(rows)} kernels")
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = name[:90]
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls {int(r['Calls']):5d} "
          f"avg {float(r['AverageNs']) / 1e3:9.1f} us  {name}")
