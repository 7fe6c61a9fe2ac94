export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out/r06u
( timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06u/gpu_tests.log 2>&1; tail -3 gpurun_out/r06u/gpu_tests.log
for i in 1 2; do python tools/train_time.py 2>/dev/null | tail -1 | cut -c1-140; COPONERF_SORT_RAYS=0 python tools/train_time.py 2>/dev/null | tail -1 | cut -c1-140; done
