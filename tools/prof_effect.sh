# does the untimed profile pass right before the timed region slow the timed steps?  (default 10 steps / 3 warm-up)
for r in 1 2 3; do
  for f in "" "--no-prof"; do
    python bench.py --no-secondary --no-cpu-baseline --no-general-leg $f > gpurun_out/pe.json 2>/dev/null
    echo "flags=[$f] $(python tools/benchsum.py gpurun_out/pe.json | cut -c1-70)"
  done
done
