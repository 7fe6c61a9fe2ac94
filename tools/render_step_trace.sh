#!/bin/bash
# One steady-state step of the headline render loop (one stream) as a kernel list: per-kernel sums and the individual launches.
#   tools/render_step_trace.sh <out dir under gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/render_step}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o r -- python "$ROOT/bench.py" --no-image --no-ref-loop --no-two-stream-pass --cpu-rays 0 --train-steps 0 --steps 5 ) > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.log"
T=$(find "$OUT/prof" -name '*kernel_trace.csv' | head -1)
python "$ROOT/tools/summarize_pmc.py" "$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)" "$OUT/render_kernel_stats.summary.csv" 40
python "$ROOT/tools/trace_step.py" "$T" project_rays 30 > "$OUT/step_kernels.txt"
python "$ROOT/tools/trace_step.py" "$T" project_rays 60 --launches > "$OUT/step_launches.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/step_kernels.txt"
