export TMPDIR=/tmp PYTHONPATH=$PWD
for i in 1 2 3 4; do python -m pytest tests/test_gpu_step.py -x -q -s -k trunk_fp16 2>&1 | grep "trunk gradients\|passed\|failed"; done
