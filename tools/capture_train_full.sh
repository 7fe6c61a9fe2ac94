#!/bin/bash
# Full kernel list (all names) and the time-ordered launch list of ONE steady-state training step, plus the library-op table.
#   tools/capture_train_full.sh <tag>
set -u
TAG=${1:-r06tf}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd "$ROOT"
python tools/train_time.py --steps 5 > "$OUT/train_time.log" 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_prof" -o t -- python "$ROOT/tools/train_time.py" --steps 3 ) > "$OUT/train_prof.log" 2>&1
T=$(find "$OUT/train_prof" -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py "$T" project_rays 1000 > "$OUT/train_step_kernels.txt" 2>&1
python tools/trace_order.py "$T" project_rays > "$OUT/train_order.txt" 2>&1
python tools/aten_time.py --top 200 > "$OUT/aten_train.txt" 2>&1
rm -rf "$OUT/train_prof"
ls "$OUT"
