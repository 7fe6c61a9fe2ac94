#!/usr/bin/env python
"""Print the rate / time keys of a bench.py JSON line (file argument), one per line."""
import json
import sys

line = [l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
for k, v in d.items():
    if "ms_per_step" in k or "rays_per_s" in k or k == "value":
        print(f"{k:42s} {v}")
