#!/bin/bash
# Round numbers: default bench (with CPU baseline) + the other workloads; JSON lines land in gpurun_out/final/.
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_gowalla.json 2> gpurun_out/final/bench_gowalla.err
python bench.py --workload ml10m --no-cpu-baseline > gpurun_out/final/bench_ml10m.json 2>/dev/null
python bench.py --workload addressa --no-cpu-baseline > gpurun_out/final/bench_addressa.json 2>/dev/null
python bench.py --train normalbce --no-cpu-baseline > gpurun_out/final/bench_gowalla_normalbce.json 2>/dev/null
python bench.py --no-defer --no-cpu-baseline > gpurun_out/final/bench_gowalla_nodefer.json 2>/dev/null
python tools/bench_lgcn.py > gpurun_out/final/bench_lightgcn_yelp2018.json 2>/dev/null
tail -c 600 gpurun_out/final/bench_gowalla.json
