export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out/r06v
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_step.py tests/test_gpu_converge.py -x -q ) > gpurun_out/r06v/tests.log 2>&1; tail -2 gpurun_out/r06v/tests.log
python tools/sort_rays_ab.py 2>/dev/null | tail -1
python tools/train_time.py --steps 8 2>/dev/null | tail -1 | cut -c1-150
