#!/bin/bash
# kernel timeline of the fused chain-shared sweep at a low chain count (chain-minor forced): usage  bash tools/timeline_fused_low.sh <chains>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-8}
OUT=$R/gpurun_out/tl_fused_$C
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp AUXSSM_CM=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --chains $C --no-secondary --no-cpu-baseline --no-general-leg --no-prof --steps 6 --warmup 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 $R/tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) k_fs_accept > $R/gpurun_out/tl_fused_$C.txt
cat $R/gpurun_out/tl_fused_$C.txt
