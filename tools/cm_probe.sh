for c in 2 4 8 16 32; do
  for cm in 0 1; do
    AUXSSM_CM=$cm timeout -k 10 200 python bench.py --chains $c --no-secondary --no-cpu-baseline --no-general-leg --steps 20 --warmup 3 > gpurun_out/cm_${c}_$cm.log 2>&1 || { tail -3 gpurun_out/cm_${c}_$cm.log; exit 1; }
    python - <<PY
import json
for l in open("gpurun_out/cm_${c}_$cm.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("chains $c cm=$cm:", round(d["value"], 1), "sweeps/s", round(d["ms_per_step"], 3), "ms/step", d["config"].get("model_sharing")[:40])
PY
  done
done
