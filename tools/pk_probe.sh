# packed-lane variant of the fused passes at few chains (AUXSSM_FS_PACK: 0 off, 1 on, n > 1: cap) x chunk length
for c in ${CS:-8 16 32}; do
  for pk in 0 1; do
    for e in ${ES:-8 16}; do
      AUXSSM_FS_PACK=$pk AUXSSM_FS_E=$e timeout -k 10 200 python tools/lowchain_probe.py c2_$c 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('chains $c pack=$pk E=$e:', d['sweeps_per_s'], 'sweeps/s', d['ms_per_sweep_call'], 'ms')"
    done
  done
done
