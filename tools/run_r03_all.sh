set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/bench_r3a.log 2>&1 || { tail -5 gpurun_out/bench_r3a.log; exit 1; }
python tools/benchsum.py gpurun_out/bench_r3a.log
bash tools/pmc_sq_c3.sh c 4096 256 > gpurun_out/sqc3_c.log 2>&1 || true
python tools/sq_summary.py gpurun_out/sqc3_c/summary.txt 4096 256 1024 --out gpurun_out/r03_sq_tmp.json
bash tools/profile_r03.sh b c3 > gpurun_out/prof3_b.log 2>&1 || { tail -5 gpurun_out/prof3_b.log; exit 1; }
tail -2 gpurun_out/prof3_b.log
