#!/bin/bash
# Capture the round's evidence on the GPU box into gpurun_out/<tag>/ (copy what should be judged into profiles/):
#   bench json (render + train block), rocprofv3 kernel stats of the render bench, PMC passes of the dominant kernel,
#   steady-state kernel lists of one get_z call and of one training step.
#   tools/capture_profiles.sh <tag>
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd "$ROOT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render_prof" -o r -- python "$ROOT/bench.py" --no-image --no-ref-loop --no-two-stream-pass --no-fresh-pair --no-f32 --cpu-rays 0 --train-steps 0 --steps 5 ) > "$OUT/bench_under_rocprof.json" 2> "$OUT/render_prof.log"
python tools/summarize_pmc.py "$(find "$OUT/render_prof" -name '*kernel_stats.csv' | head -1)" "$OUT/render_kernel_stats.summary.csv" 40
# sidecar of the kernel statistics: which kernel source / launch shape they were taken on (bench.py roofline.rocprof)
python - "$OUT" <<'PY'
import hashlib, json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
src = os.path.join(root, "coponerf_amd", "csrc", "encode_fused.hip")
json.dump({"kernel": "encode_fused_kernel",
           "kernel_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "rows_per_launch": 16777216,
           "command": "python bench.py --no-image --no-ref-loop --no-two-stream-pass --no-fresh-pair --no-f32 --cpu-rays 0 --train-steps 0 --steps 5 (under rocprofv3 --kernel-trace --stats)"},
          open(os.path.join(sys.argv[1], "render_kernel_stats.meta.json"), "w"), indent=1)
PY
# PMC passes (one counter group per run) of the dominant kernel on the headline command itself: 65 536-ray launches
tools/pmc_passes.sh "$OUT/pmc_encode" encode_fused -- python "$ROOT/bench.py" --no-image --no-ref-loop --no-two-stream-pass --no-fresh-pair --no-f32 --cpu-rays 0 --train-steps 0 --steps 3 --warmup 1 > "$OUT/pmc_encode.log" 2>&1
python tools/make_traffic_json.py "$OUT/pmc_encode/summary.json" 16777216 "$OUT/traffic.json" encode_key >> "$OUT/pmc_encode.log" 2>&1
# the reference-arithmetic mode: per-kernel statistics of one image (one stream)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f32_prof" -o f -- python "$ROOT/tools/f32_time.py" --steps 2 ) > "$OUT/f32_time.log" 2>&1
python tools/summarize_pmc.py "$(find "$OUT/f32_prof" -name '*kernel_stats.csv' | head -1)" "$OUT/f32_kernel_stats.summary.csv" 20
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/getz_prof" -o g -- python "$ROOT/tools/getz_time.py" ) > "$OUT/getz_prof.log" 2>&1
python tools/trace_step.py "$(find "$OUT/getz_prof" -name '*kernel_trace.csv' | head -1)" soft_argmax_cols 45 > "$OUT/getz_step_kernels.txt" 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_prof" -o t -- python "$ROOT/tools/train_time.py" --steps 3 ) > "$OUT/train_prof.log" 2>&1
python tools/trace_step.py "$(find "$OUT/train_prof" -name '*kernel_trace.csv' | head -1)" project_rays 60 > "$OUT/train_step_kernels.txt" 2>&1
# the step's own line from an UNPROFILED run (under rocprofv3 the host's per-launch cost makes the step host-bound: 105 ms)
python tools/train_time.py --steps 5 2>/dev/null | grep -h train_ms_per_step > "$OUT/train_step.json"
python tools/aten_time.py --top 120 > "$OUT/aten_train.txt" 2>&1
python tools/wgrad_f32_bench.py > "$OUT/wgrad_f32_bench.json" 2>/dev/null
python tools/trunk_conv_bench.py > "$OUT/trunk_conv_bench.json" 2>/dev/null
python tools/trace_order.py "$(find "$OUT/getz_prof" -name '*kernel_trace.csv' | head -1)" soft_argmax_cols > "$OUT/getz_order.txt" 2>&1
bash tools/refloop_trace.sh "gpurun_out/$TAG/ref_loop_b1" 1 > /dev/null 2>&1
bash tools/refloop_trace.sh "gpurun_out/$TAG/ref_loop_b2" 2 > /dev/null 2>&1
python tools/refloop_profile.py --batch 1 --top 8 > "$OUT/ref_loop_b1_host.txt" 2>&1
python tools/getz_graph.py > "$OUT/getz_graph.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.csv" -size +2M -delete
ls "$OUT"
