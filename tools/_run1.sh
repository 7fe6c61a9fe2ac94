export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06h
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06h/gpu_tests.log 2>&1; tail -6 gpurun_out/r06h/gpu_tests.log
