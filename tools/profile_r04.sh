#!/bin/bash
# Round-4 profile passes (same passes as profile_r04.sh; output prefix prof4_) on the GPU box (rocprofv3 kernel stats and PMC counters in SEPARATE passes, as the guide prescribes).
# usage: bash tools/profile_r04.sh <tag> [c2|c3|c5 ...]     -> gpurun_out/prof4_<tag>/{stats_*,pmc_*}/...csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof4_${1:-x}
shift || true
WHAT=${@:-c2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C2="$R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-general-leg"
C2G="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-share-model"
C5="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c5 --T 1024 --chains 4"
C3="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c3 --T 1024 --chains 4"
run() { name=$1; shift; echo "== $name"; timeout -k 10 500 rocprofv3 "$@" > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }; }
for w in $WHAT; do
  case $w in
    c2) run stats_c2 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $C2
        run pmc_fetch_c2 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2 -- python3 $C2
        run pmc_write_c2 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2 -- python3 $C2 ;;
    c2g) run stats_c2g --kernel-trace --stats --output-format csv -d $OUT/stats_c2g -- python3 $C2G
        run pmc_fetch_c2g --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c2g -- python3 $C2G
        run pmc_write_c2g --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c2g -- python3 $C2G ;;
    c5) run stats_c5 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python3 $C5
        run pmc_mfma_c5 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma_c5 -- python3 $C5 ;;
    c3) run stats_c3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $C3
        run pmc_fetch_c3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c3 -- python3 $C3
        run pmc_write_c3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c3 -- python3 $C3 ;;
  esac
done
find $OUT -name "*.csv" | head -40
du -sh $OUT
