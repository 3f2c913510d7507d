#!/bin/bash
# SQ-level counters of the cSMC sweep kernels (k_csmc_fwd / k_csmc_bwd) on C3's shape (SV, N = 1024, backward sampling), each counter set in its own
# rocprofv3 --pmc pass, program directly after `--`.  usage: bash tools/pmc_sq_c3.sh <tag> [T] [chains]   -> gpurun_out/sqc3_<tag>/p*/...csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sqc3_${1:-x}
T=${2:-4096}
CH=${3:-256}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --workload csmc --T $T --chains $CH"
i=0
failed=""
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $CMD > $OUT/p$i.log 2>&1 || { tail -3 $OUT/p$i.log; failed="$failed p$i"; }
done
echo "failed passes:${failed:- none}"
for f in $(find $OUT -name "*counter_collection.csv"); do python3 $R/tools/pmc_raw.py $f; done > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
