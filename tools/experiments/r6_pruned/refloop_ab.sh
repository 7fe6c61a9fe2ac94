#!/bin/bash
# The reference callers' 18-call loop (bench.py ref_loop block) under engine switches, in alternation on one box:
#   tools/refloop_ab.sh   -> one line per run: switch, single-call ms, batch-1 loop ms (min, max), batch-2 loop ms
run() {
  env "$@" python bench.py --steps 10 --warmup 3 --train-steps 0 --no-image --no-f32 --cpu-rays 0 --no-two-stream-pass --no-fresh-pair > gpurun_out/rl.json 2>/dev/null
  python - "$*" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/rl.json').read().splitlines() if l.startswith('{')][-1])
r = d['ref_loop']
print("%-28s single %.2f  b1 %.2f (%.2f .. %.2f)  b2 %.2f" % (sys.argv[1], d['ms_per_step'], r['batch1']['render_ms_per_batch'],
      *r['batch1']['render_ms_min_max'], r['batch2']['render_ms_per_batch']))
PY
}
mkdir -p gpurun_out
for i in 1 2; do
  run X=1
  run COPONERF_CE_RECOMPUTE=0
  run COPONERF_UNIT_ORDER=0
done
