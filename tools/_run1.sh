export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06g
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06g/gpu_tests.log 2>&1; tail -8 gpurun_out/r06g/gpu_tests.log
