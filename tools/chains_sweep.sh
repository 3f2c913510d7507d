for c in ${CHAINS:-1 8 64}; do
  timeout -k 10 300 python bench.py --chains $c --no-secondary --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ch_$c.log 2>&1 || { tail -3 gpurun_out/ch_$c.log; exit 1; }
  python - <<PY
import json
for l in open("gpurun_out/ch_$c.log"):
    if l.startswith("{"):
        d = json.loads(l); g = d.get("general_path")
        print("chains $c:", round(d["value"], 1), "sweeps/s", round(d["ms_per_step"], 3), "ms/step", d["config"].get("model_sharing"), "| general:", round(g["value"], 1) if g else None)
PY
done
