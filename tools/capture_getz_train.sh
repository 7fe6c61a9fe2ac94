#!/bin/bash
# rocprofv3 kernel traces of one get_z call and one training step (steady state): per-kernel sums (trace_step.py) and the
# time-ordered launch list of get_z (trace_order.py).   tools/capture_getz_train.sh <tag>
set -u
TAG=${1:-r03gt}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd "$ROOT"
python tools/getz_time.py > "$OUT/getz_time.log" 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/getz_prof" -o g -- python "$ROOT/tools/getz_time.py" ) > "$OUT/getz_prof.log" 2>&1
G=$(find "$OUT/getz_prof" -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py "$G" soft_argmax_cols 60 > "$OUT/getz_step_kernels.txt" 2>&1
python tools/trace_order.py "$G" soft_argmax_cols > "$OUT/getz_order.txt" 2>&1
if [ "${2:-}" != "nogz" ]; then
python tools/train_time.py --steps 3 > "$OUT/train_time.log" 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_prof" -o t -- python "$ROOT/tools/train_time.py" --steps 3 ) > "$OUT/train_prof.log" 2>&1
T=$(find "$OUT/train_prof" -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py "$T" project_rays 70 > "$OUT/train_step_kernels.txt" 2>&1
fi
rm -rf "$OUT/getz_prof" "$OUT/train_prof"
ls "$OUT"
