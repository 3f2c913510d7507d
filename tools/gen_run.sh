timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/g.log 2>&1 || { tail -3 gpurun_out/g.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/g.log"):
    if l.startswith("{"):
        d = json.loads(l); g = d["general_path"]
        print("headline", round(d["value"]), {k: v["ms_per_step"] for k, v in d["kernels"].items()})
        print("general", round(g["value"]), g["roofline"]["frac"], {k: v["ms_per_step"] for k, v in g["kernels"].items()})
PY
