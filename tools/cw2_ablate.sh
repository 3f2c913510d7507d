#!/bin/bash
# phase-ablation timings of the wide-state cSMC kernels (variants built with -DCW2_ABL=<mask> as aux_ssm_samplers_amd/libauxssm_cw<mask>.so): SV protocol, 1 and 256 chains
cd ${GRAFT_REPO_ROOT:-.}
for k in 1 2 4 8 16 31; do echo "mask $k"; AUXSSM_LIB=$PWD/aux_ssm_samplers_amd/libauxssm_cw$k.so timeout -k 10 120 python tools/bench_configs.py sv30 2>&1 | grep -o '"chains": [0-9]*, "sweeps_per_s": [0-9.]*, "ms_per_sweep_call": [0-9.]*' | head -2; done
echo full; timeout -k 10 120 python tools/bench_configs.py sv30 2>&1 | grep -o '"chains": [0-9]*, "sweeps_per_s": [0-9.]*, "ms_per_sweep_call": [0-9.]*' | head -2
