#!/bin/bash
# rocprofv3 counter passes (one --pmc group per run: SQ 8 / TCC 4 slots; never combined with tracing) for one command.
#   tools/pmc_passes.sh <out_dir> <kernel_substring[;substring...]> -- <command...>
# Writes <out_dir>/pass<i>/ (raw CSVs) and <out_dir>/summary.json (mean counter value per dispatch of the kernel).
set -u
OUT=$(realpath -m "$1"); KSUB="$2"; shift 3
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC_GROUPS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_ANY"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE"
)
i=0
for g in "${PMC_GROUPS[@]}"; do
  d="$OUT/pass$i"; mkdir -p "$d"
  ( cd /tmp && rocprofv3 --pmc $g --output-format csv -d "$d" -- "$@" ) > "$d/run.log" 2>&1 || echo "pass $i failed" >&2
  i=$((i+1))
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, json, os, re, sys, collections
out, ksubs = sys.argv[1], sys.argv[2].split(";")          # several kernels: ";"-separated substrings (kernel names contain commas)
aggs = [collections.OrderedDict() for _ in ksubs]
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        for ksub, agg in zip(ksubs, aggs):
            if ksub in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for i, (ksub, agg) in enumerate(zip(ksubs, aggs)):
    res = {k: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for k, v in agg.items()}
    name = "summary.json" if i == 0 else "summary_" + re.sub(r"[^A-Za-z0-9]+", "_", ksub).strip("_") + ".json"
    json.dump({"kernel": ksub, "counters": res}, open(os.path.join(out, name), "w"), indent=1)
    print(ksub, json.dumps({k: round(v["mean_per_dispatch"], 1) for k, v in res.items()}))
PY
# keep only the summaries and logs (raw CSVs are large)
find "$OUT" -name "*.csv" -size +2M -delete
