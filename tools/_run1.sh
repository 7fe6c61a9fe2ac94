export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06z
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06z/gpu_tests.log 2>&1; tail -6 gpurun_out/r06z/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06z/smoke.log 2>&1; tail -2 gpurun_out/r06z/smoke.log
python bench.py --mode train > gpurun_out/r06z/bench_train.json 2> gpurun_out/r06z/bench_train.err; python tools/show_rates.py gpurun_out/r06z/bench_train.json | head -5
