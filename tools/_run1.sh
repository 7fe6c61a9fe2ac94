export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06d
echo "== gather + GEMM form (no tables)"; python tools/grad_floor_probe.py --rays 1024 --hip --no-tables 2>&1 | grep HIP
echo "== tables forward, gather-form backward"; COPONERF_TABLE_BACKWARD=0 python tools/grad_floor_probe.py --rays 1024 --hip 2>&1 | grep HIP
echo "== scale target 4096"; python - <<'PY'
import coponerf_amd.render as r
print("default grad_scale_target", r.RenderEngine().grad_scale_target)
PY
