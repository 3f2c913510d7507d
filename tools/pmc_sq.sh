#!/bin/bash
# SQ-level counters of the C2 headline kernels (issue vs wait): separate rocprofv3 --pmc passes.  usage: bash tools/pmc_sq.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C2="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-general-leg --no-prof"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $C2 > $OUT/p$i.log 2>&1 || { tail -3 $OUT/p$i.log; exit 1; }
done
find $OUT -name "*counter_collection.csv" | head
