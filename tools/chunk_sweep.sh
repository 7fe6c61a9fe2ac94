for c in 512 1024 2048 4096 8192 16384; do
  python bench.py --chunk-rays $c --steps 5 --warmup 2 --train-steps 0 --no-image --no-ref-loop --no-f32 --cpu-rays 0 --no-two-stream-pass > gpurun_out/chunk_$c.json 2>/dev/null
  echo "chunk $c: $(python tools/show_rates.py gpurun_out/chunk_$c.json | grep '^ms_per_step ')"
done
