for r in 1 2 3; do
for v in base stage4; do
  AUXSSM_LIB=$PWD/aux_ssm_samplers_amd/libauxssm_$v.so python bench.py --no-secondary --no-cpu-baseline --no-general-leg --steps 30 --warmup 5 > gpurun_out/ab_$v.json 2>/dev/null
  echo "$v: $(python tools/benchsum.py gpurun_out/ab_$v.json | cut -c1-140)"
done
done
