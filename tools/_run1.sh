export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
for mode in cpl4 prod; do
OUT=$ROOT/gpurun_out/r06y_$mode
mkdir -p $OUT
LIB=""; [ $mode = cpl4 ] && LIB=$ROOT/tools/_build/libcpn_bucket_cpl4.so
( cd /tmp && COPONERF_HIP_LIB=$LIB rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_prof" -o t -- python "$ROOT/tools/train_time.py" --steps 3 ) > "$OUT/train_prof.log" 2>&1
T=$(find "$OUT/train_prof" -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py "$T" project_rays 1000 > "$OUT/train_step_kernels.txt" 2>&1
rm -rf "$OUT/train_prof"
grep "bucket_accumulate" $OUT/train_step_kernels.txt | cut -c1-80
done
