#!/bin/bash
# Round 6: what the layout crossings of cpn_encode_key cost, measured — timing-only builds of the kernel (results wrong)
# with the K = 80 accumulators left in the MFMA layout (2048), the B operand of the key layer not crossed (4096), both
# (6144): the most ANY replacement of the ds_bpermute crossings (LDS transposes included) could give back.
#   tools/encode_cross_ablate.sh build      (here: cross-compiles the variants into tools/_build/)
#   tools/encode_cross_ablate.sh run <out>  (on the GPU box: the one-stream headline loop with each library)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CS=$ROOT/coponerf_amd/csrc
B=$ROOT/tools/_build
if [ "$1" = build ]; then
  mkdir -p "$B"
  OBJS=$(ls $CS/*.o | grep -v encode_fused.o)
  for A in 0 2048 4096 6144; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DCPN_EF_ABLATE=$A -x hip -c $CS/encode_fused.hip -o $B/ef_$A.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $B/ef_$A.o -o $B/libcpn_ef_$A.so || exit 1
  done
  ls -la $B/libcpn_ef_*.so
else
  OUT=$2; mkdir -p "$OUT"; export PYTHONPATH=$ROOT TMPDIR=/tmp
  for rep in 1 2; do for A in 0 2048 4096 6144; do
    COPONERF_HIP_LIB=$B/libcpn_ef_$A.so python $ROOT/bench.py --no-image --no-ref-loop --no-two-stream-pass --no-fresh-pair --no-f32 --cpu-rays 0 --train-steps 0 --steps 5 > "$OUT/ef_${A}_$rep.json" 2> "$OUT/ef_${A}_$rep.err"
    python - "$OUT/ef_${A}_$rep.json" $A <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ablate", sys.argv[2], "encode_key ms", round(d["kernel_breakdown"]["encode_key"]["ms_per_step"], 3), "step ms", round(d["ms_per_step"], 3))
PY
  done; done
fi
