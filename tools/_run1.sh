export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_range.py -x -q -s -k "peaked or sharpness" > gpurun_out/r06e/peaked.log 2>&1; grep -E "peaked attention:|passed|failed|Error|^ +[0-9]+ +[0-9.]+ +|gain" gpurun_out/r06e/peaked.log | head -30
