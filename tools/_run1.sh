export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06l
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06l/gpu_tests.log 2>&1; tail -6 gpurun_out/r06l/gpu_tests.log
python bench.py > gpurun_out/r06l/bench.json 2> gpurun_out/r06l/bench.err; python tools/show_rates.py gpurun_out/r06l/bench.json | head -40
