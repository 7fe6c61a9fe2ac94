export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_train.py tests/test_gpu_step.py -x -q > gpurun_out/r06b/tests.log 2>&1; tail -5 gpurun_out/r06b/tests.log
python tools/train_time.py --steps 5 > gpurun_out/r06b/train_time.log 2>&1; tail -1 gpurun_out/r06b/train_time.log
COPONERF_TRAIN_FUSE_KEY=0 python tools/train_time.py --steps 5 > gpurun_out/r06b/train_time_nofusekey.log 2>&1; tail -1 gpurun_out/r06b/train_time_nofusekey.log
python tools/launch_sites.py 80 > gpurun_out/r06b/launch_sites.txt 2>&1
