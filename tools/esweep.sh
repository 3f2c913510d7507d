for e in 96 128 192 256 384; do
  AUXSSM_SCAN_E=$e timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/es_$e.log 2>&1 || exit 1
  python - <<PY
import json
for l in open("gpurun_out/es_$e.log"):
    if l.startswith("{"):
        d = json.loads(l); g = d["general_path"]
        print("E=$e headline", round(d["value"]), "general", round(g["value"]), {k: v["ms_per_step"] for k, v in g["kernels"].items() if k in ("filter_scan", "sample_scan")})
PY
done
