export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_train.py -x -q -k "combine or table_form or oracle_autograd or end_to_end" > gpurun_out/r06a/tests.log 2>&1; tail -5 gpurun_out/r06a/tests.log
python tools/train_time.py --steps 5 > gpurun_out/r06a/train_time.log 2>&1; tail -2 gpurun_out/r06a/train_time.log
COPONERF_FUSE_COMBINE=0 python tools/train_time.py --steps 5 > gpurun_out/r06a/train_time_nofuse.log 2>&1; tail -1 gpurun_out/r06a/train_time_nofuse.log
python tools/aten_time.py --top 400 --kernels "elementwise_kernel_manual_unroll|CUDAFunctor_add|copyBuffer|FillFunctor|SubTensorOp" > gpurun_out/r06a/aten_elementwise.txt 2>&1
