#!/bin/bash
# phase-ablation timings of wk_gain_tab (variants built with -DAUXSSM_GT_PHASE=1..5 as aux_ssm_samplers_amd/libauxssm_gt<k>.so)
cd ${GRAFT_REPO_ROOT:-.}
for k in 1 2 3 4 5; do AUXSSM_LIB=$PWD/aux_ssm_samplers_amd/libauxssm_gt$k.so timeout -k 10 120 python tools/c5_shared_probe.py 16 3 || exit 1; done
timeout -k 10 120 python tools/c5_shared_probe.py 16 3
