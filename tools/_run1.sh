export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
OUT=$ROOT/gpurun_out/r06v
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_step.py -x -q ) > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_prof" -o t -- python "$ROOT/tools/train_time.py" --steps 3 ) > "$OUT/train_prof.log" 2>&1
T=$(find "$OUT/train_prof" -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py "$T" project_rays 1000 > "$OUT/train_step_kernels.txt" 2>&1
rm -rf "$OUT/train_prof"
head -1 $OUT/train_step_kernels.txt
grep "node_features" $OUT/train_step_kernels.txt | cut -c1-100
