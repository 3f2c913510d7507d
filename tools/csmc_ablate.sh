#!/bin/bash
# Diagnostic: build libauxssm_abl<mask>.so variants of the cSMC unit with one forward-step phase removed (csmc.hip: AUXSSM_CSMC_ABLATE) -- CPU side.
# usage: bash tools/csmc_ablate.sh build "1 2 4 8 16"   (here);   bash tools/csmc_ablate.sh run "1 2 4 8 16"   (GPU box)
set -e
cd "$(dirname "$0")/../aux_ssm_samplers_amd/csrc"
if [ "$1" = build ]; then
  for m in $2; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DAUXSSM_CSMC_ABLATE=$m -c csmc.hip -o /tmp/csmc_abl$m.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 api.o /tmp/csmc_abl$m.o pit.o loop.o wide.o inst_*.o -o ../libauxssm_abl$m.so
  done
else
  cd ../..
  for m in $2; do
    AUXSSM_LIB=$PWD/aux_ssm_samplers_amd/libauxssm_abl$m.so timeout -k 10 200 python bench.py --secondary c3 --no-cpu-baseline --steps 2 --warmup 1 --T 1024 --chains 4 > gpurun_out/c3_abl$m.log 2>&1 || { tail -3 gpurun_out/c3_abl$m.log; exit 1; }
    python - <<PY
import json
for l in open("gpurun_out/c3_abl$m.log"):
    if l.startswith("{"):
        d = json.loads(l)["secondary"]["C3_csmc"]
        print("ablate $m:", d["value"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
  done
fi
