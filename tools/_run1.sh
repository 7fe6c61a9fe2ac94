export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_dist.py tests/test_gpu_train.py -x -q > gpurun_out/r06c/tests.log 2>&1; tail -3 gpurun_out/r06c/tests.log
python bench.py --mode train --steps 6 --warmup 2 > gpurun_out/r06c/bench_train.json 2> gpurun_out/r06c/bench_train.err; tail -c 1500 gpurun_out/r06c/bench_train.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --mode train --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06c/bench_train2.json 2> gpurun_out/r06c/bench_train2.err; tail -c 600 gpurun_out/r06c/bench_train2.err; tail -c 2500 gpurun_out/r06c/bench_train2.json
