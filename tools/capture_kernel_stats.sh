#!/bin/bash
# Only the rocprofv3 kernel statistics of the render bench (+ their sidecar) of tools/capture_profiles.sh:
#   tools/capture_kernel_stats.sh <tag>   -> gpurun_out/<tag>/{bench_under_rocprof.json,render_kernel_stats.summary.csv,render_kernel_stats.meta.json}
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd "$ROOT"
CMD="--no-image --no-ref-loop --no-two-stream-pass --no-fresh-pair --no-f32 --cpu-rays 0 --train-steps 0 --steps 5"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render_prof" -o r -- python "$ROOT/bench.py" $CMD ) > "$OUT/bench_under_rocprof.json" 2> "$OUT/render_prof.log"
python tools/summarize_pmc.py "$(find "$OUT/render_prof" -name '*kernel_stats.csv' | head -1)" "$OUT/render_kernel_stats.summary.csv" 40
python - "$OUT" "$CMD" <<'PY'
import hashlib, json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
src = os.path.join(root, "coponerf_amd", "csrc", "encode_fused.hip")
json.dump({"kernel": "encode_fused_kernel", "kernel_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
           "rows_per_launch": 16777216, "command": f"python bench.py {sys.argv[2]} (under rocprofv3 --kernel-trace --stats)"},
          open(os.path.join(sys.argv[1], "render_kernel_stats.meta.json"), "w"), indent=1)
PY
find "$OUT" -name "*kernel_trace.csv" -delete
rm -rf "$OUT/render_prof"
cat "$OUT/render_kernel_stats.summary.csv" | head -20
