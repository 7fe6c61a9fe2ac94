#!/usr/bin/env python
"""ms_per_step and the per-kernel HIP-event times (kernel_breakdown) of a bench.py JSON line (file argument)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("ms_per_step %.3f" % d["ms_per_step"], " ".join("%s=%.3f" % (k, v["ms_per_step"]) for k, v in d.get("kernel_breakdown", {}).items()))
