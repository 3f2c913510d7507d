timeout -k 10 300 python tools/bench_configs.py c4 > gpurun_out/c4.log 2>&1 || { tail -5 gpurun_out/c4.log; exit 1; }
cat gpurun_out/c4.log
