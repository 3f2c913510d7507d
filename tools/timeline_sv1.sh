#!/bin/bash
# kernel timeline of the SV first-order (chain-shared) aux-Kalman sweep: bash tools/timeline_sv1.sh <chains>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-1024}
OUT=$R/gpurun_out/tl_sv1_$C
mkdir -p $OUT
cd $R && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/c3k_probe.py 1 $C 6 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) k_sv_accept > gpurun_out/tl_sv1_$C.txt
cat gpurun_out/tl_sv1_$C.txt
