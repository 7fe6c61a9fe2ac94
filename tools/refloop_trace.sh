#!/bin/bash
# rocprofv3 kernel trace of the reference callers' 18-call loop (bench.py ref_loop): wall / busy / per-kernel sums of the
# last loop and the launch timeline of one call (tools/refloop_trace.py).
#   bash tools/refloop_trace.sh <outdir under the repo> [batch]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/${1:-gpurun_out/refloop_trace}
mkdir -p "$out"
export TMPDIR=/tmp PYTHONPATH=$ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -o loop -- python "$ROOT/tools/refloop_profile.py" --batch ${2:-1} --trace-only ) > "$out/run.log" 2>&1
python "$ROOT/tools/refloop_trace.py" "$(find "$out/raw" -name '*kernel_trace.csv' | head -1)" > "$out/loop.txt" 2>&1
rm -rf "$out/raw"
