export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06j
python tools/train_time.py > gpurun_out/r06j/train_time.log 2>&1; tail -1 gpurun_out/r06j/train_time.log
CPN_TRUNK_BWD_F16=0 python tools/train_time.py > gpurun_out/r06j/train_time_f32bwd.log 2>&1; tail -1 gpurun_out/r06j/train_time_f32bwd.log
python tools/host_step_profile.py --ops > gpurun_out/r06j/host_ops_f16.txt 2>&1; grep "^host:" gpurun_out/r06j/host_ops_f16.txt; sed -n 6,14p gpurun_out/r06j/host_ops_f16.txt | cut -c1-150
