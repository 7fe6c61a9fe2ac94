#!/bin/bash
# Why does cpn_encode_hidden take 1.7 ms back to back and 2.1-2.3 ms inside the chunk loop?  rocprofv3 counter passes
# (one --pmc group per run, nothing else enabled) over the SAME launch in two regimes: hot (launches back to back) and
# with 1 GiB of unrelated traffic between launches (tools/encode_ablate.py --flush: the in-loop time reproduces).
#   tools/pmc_gap.sh <out_dir>     -> <out_dir>/gap.json: mean counter per launch in both regimes (last 4 launches of each run)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
GROUPS_=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"
 "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
for regime in hot flush; do
  extra=""; [ "$regime" = flush ] && extra="--flush --flush-mb 512"
  i=0
  for g in "${GROUPS_[@]}"; do
    d="$OUT/$regime/pass$i"; mkdir -p "$d"
    ( cd /tmp && rocprofv3 --pmc $g --output-format csv -d "$d" -- python "$ROOT/tools/encode_ablate.py" --only "mt1 w16 0:" --iters 4 $extra ) > "$d/run.log" 2>&1 || echo "$regime pass $i failed" >&2
    i=$((i+1))
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = {}
for regime in ("hot", "flush"):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(out, regime, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "encode_hidden" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in per.items():
            v.sort()
            last = [x for _, x in v[-4:]]
            agg[k] = sum(last) / len(last)
    res[regime] = agg
res["ratio_flush_over_hot"] = {k: (res["flush"][k] / res["hot"][k] if res["hot"].get(k) else None) for k in res["flush"]}
json.dump(res, open(os.path.join(out, "gap.json"), "w"), indent=1)
for k in res["flush"]:
    print(f"{k:45s} hot {res['hot'].get(k, 0):16.1f}  flush {res['flush'][k]:16.1f}  x{res['ratio_flush_over_hot'][k] or 0:.2f}")
PY
find "$OUT" -name "*.csv" -delete
find "$OUT" -name "*.db" -delete
