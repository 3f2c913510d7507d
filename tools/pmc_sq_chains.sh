#!/bin/bash
# SQ counters of the fused passes at a given chain count: bash tools/pmc_sq_chains.sh <tag> <chains>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sqch_${1:-x}
C=${2:-64}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --chains $C --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-general-leg --no-prof"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $CMD > $OUT/p$i.log 2>&1 || { tail -3 $OUT/p$i.log; exit 1; }
done
for f in $(find $OUT -name "*counter_collection.csv"); do python3 $R/tools/pmc_raw.py $f; done 2>/dev/null | grep -E "k_fs_ac|k_fs_e<" > $OUT/summary.txt
cat $OUT/summary.txt
