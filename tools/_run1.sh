export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out/r06q
( timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06q/gpu_tests.log 2>&1; tail -3 gpurun_out/r06q/gpu_tests.log
