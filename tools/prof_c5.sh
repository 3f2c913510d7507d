#!/bin/bash
# rocprofv3 kernel stats of the C5 leg (d = p = 64, T = 8192, 1 and 16 sequences per launch) -> gpurun_out/prof_c5_<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_c5_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c5 --no-general-leg > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 $R/tools/kstats.py $F 2>/dev/null | grep -i "wk_" | head -40 || grep "wk_" $F | head -40
