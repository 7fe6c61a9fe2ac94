#!/bin/bash
# BASELINE configs[3] (ACID-like wide rig, 256x256x64) and configs[4] (512x512x128, 8 pairs rendered pair by pair) on one
# MI355X: bench line, rocprofv3 kernel stats of the same command, FETCH/WRITE/TCC counters of the dominant kernel.
#   tools/capture_configs.sh <tag>      -> gpurun_out/<tag>/{c4,c5}_*
set -u
TAG=${1:-r04cfg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd "$ROOT"
C4="--rig wide --train-steps 0 --no-image --no-ref-loop --no-f32"
C5="--height 512 --samples 128 --pairs 8 --pair-by-pair --steps 2 --warmup 1 --train-steps 0 --no-image --no-ref-loop"
python bench.py $C4 --cpu-rays 4096 > "$OUT/c4_bench.json" 2> "$OUT/c4_bench.err"
python bench.py $C5 --cpu-rays 1024 > "$OUT/c5_bench.json" 2> "$OUT/c5_bench.err"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4_prof" -o r -- python "$ROOT/bench.py" $C4 --cpu-rays 0 ) > "$OUT/c4_bench_under_rocprof.json" 2> "$OUT/c4_prof.log"
python tools/summarize_pmc.py "$(find "$OUT/c4_prof" -name '*kernel_stats.csv' | head -1)" "$OUT/c4_kernel_stats.summary.csv" 30
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_prof" -o r -- python "$ROOT/bench.py" $C5 --cpu-rays 0 ) > "$OUT/c5_bench_under_rocprof.json" 2> "$OUT/c5_prof.log"
python tools/summarize_pmc.py "$(find "$OUT/c5_prof" -name '*kernel_stats.csv' | head -1)" "$OUT/c5_kernel_stats.summary.csv" 30
# FETCH / WRITE / TCC counters of the dominant kernel on the configs' own bench commands (one counter group per run)
tools/pmc_passes.sh "$OUT/c4_pmc_encode" encode_fused -- python "$ROOT/bench.py" $C4 --cpu-rays 0 --no-two-stream-pass --no-fresh-pair --steps 3 --warmup 1 > "$OUT/c4_pmc_encode.log" 2>&1
tools/pmc_passes.sh "$OUT/c5_pmc_encode" encode_fused -- python "$ROOT/bench.py" $C5 --cpu-rays 0 --no-two-stream-pass --no-fresh-pair --no-f32 > "$OUT/c5_pmc_encode.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.csv" -size +2M -delete
rm -rf "$OUT"/c4_prof "$OUT"/c5_prof
ls "$OUT"
