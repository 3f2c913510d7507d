for c in 8 16 32 64 128; do
  for l in 64 128 256; do
    AUXSSM_FS_PACK_LANES=$l timeout -k 10 200 python tools/lowchain_probe.py c2_$c 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('chains $c lanes=$l:', d['sweeps_per_s'], 'sweeps/s', d['ms_per_sweep_call'], 'ms')"
  done
done
