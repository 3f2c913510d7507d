timeout -k 10 200 python bench.py --secondary c3 --no-cpu-baseline --steps 3 --warmup 1 --T 1024 --chains 4 > gpurun_out/c3.log 2>&1 || exit 1
python - <<'PY'
import json
for l in open("gpurun_out/c3.log"):
    if l.startswith("{"):
        d = json.loads(l)["secondary"]["C3_csmc"]
        print("C3", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
