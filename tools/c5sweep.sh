for n in ${C5_SWEEP:-0 256}; do
  if [ $n = 0 ]; then unset AUXSSM_WIDE_NCHUNK; else export AUXSSM_WIDE_NCHUNK=$n; fi
  timeout -k 10 200 python bench.py --secondary c5 --no-cpu-baseline --steps 3 --warmup 1 --T 1024 --chains 4 > gpurun_out/c5_n$n.log 2>&1 || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/c5_n*.log')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            for r in d['secondary']['C5_wide']['runs']:
                print(f, r['sequences_per_launch'], r['filters_per_s'], {k: v['ms_per_step'] if isinstance(v, dict) else v for k, v in r['kernels'].items() if 'filter' in k}, r['roofline']['frac'])
PY
