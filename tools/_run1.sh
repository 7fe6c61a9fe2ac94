export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06f
python tools/f32_time.py --steps 3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_range.py tests/test_cabi.py -x -q > gpurun_out/r06f/parity.log 2>&1; tail -2 gpurun_out/r06f/parity.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-ref-loop --no-fresh-pair --train-steps 0 --no-image > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","ms_per_step","rays_per_s_f32","rgb_max_abs_f16_vs_f32","f32_mode")})
PY
