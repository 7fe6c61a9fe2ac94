export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06n
( time timeout 900 python -m pytest tests/test_gpu_step.py -x -q -s -k trunk_fp16 ) > gpurun_out/r06n/tests.log 2>&1; tail -5 gpurun_out/r06n/tests.log; grep -n "largest fp16\|trunk gradients" gpurun_out/r06n/tests.log
