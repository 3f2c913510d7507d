for c in 1 256; do for p in "" "--no-prof"; do
  timeout -k 10 300 python bench.py --chains $c --no-secondary --no-cpu-baseline --steps 10 --warmup 3 $p > gpurun_out/np.log 2>&1 || { tail -3 gpurun_out/np.log; exit 1; }
  python - <<PY
import json
for l in open("gpurun_out/np.log"):
    if l.startswith("{"):
        d = json.loads(l); print("chains $c prof='$p':", round(d["value"], 1), round(d["ms_per_step"], 4))
PY
done; done
