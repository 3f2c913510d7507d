# fused chain-minor sweep at low chain counts: chunk length sweep
for c in ${CS:-8 16 32}; do
  for e in ${ES:-16 24 32 48 64}; do
    AUXSSM_CM=1 AUXSSM_FS_E=$e timeout -k 10 200 python bench.py --chains $c --no-secondary --no-cpu-baseline --no-general-leg --no-prof --steps 30 --warmup 5 > gpurun_out/cm2_${c}_$e.log 2>&1 || { tail -3 gpurun_out/cm2_${c}_$e.log; exit 1; }
    python - <<PY
import json
for l in open("gpurun_out/cm2_${c}_$e.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("chains $c E=$e:", round(d["value"], 1), "sweeps/s", round(d["ms_per_step"], 3), "ms/step")
PY
  done
done
