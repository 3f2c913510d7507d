R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c4stats
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_configs.py c4 > $OUT/run.log 2>&1 || { tail -3 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, re, subprocess
f = glob.glob("$OUT/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
names = subprocess.run(["c++filt"], input="\n".join(r["Name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in list(zip(rows, names))[:22]:
    n = re.sub(r"\(.*$", "", re.sub(r"^void ax::", "", n))
    print(f"{n[:72]:72s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
