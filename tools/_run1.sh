export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
OUT=$ROOT/gpurun_out/r06fin8
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/gpu_tests.log 2>&1; tail -4 $OUT/gpu_tests.log
for i in 1 2; do python tools/train_time.py --steps 8 2>/dev/null | tail -1 | tee $OUT/train_step.json | cut -c1-130; done
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/train_prof" -o t -- python "$ROOT/tools/train_time.py" --steps 3 ) > "$OUT/train_prof.log" 2>&1
T=$(find "$OUT/train_prof" -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py "$T" project_rays 60 > "$OUT/train_step_kernels.txt" 2>&1
rm -rf "$OUT/train_prof"
head -1 $OUT/train_step_kernels.txt
python tools/aten_time.py --top 120 > "$OUT/aten_train.txt" 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python tools/show_rates.py $OUT/bench.json | head -3
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['train']['ms_per_step'], d['roofline']['frac'], d['get_z_ms'])"
