#!/bin/bash
# Round numbers: default bench as the driver runs it and as the README quotes it, the other workloads; JSON lines land in gpurun_out/final/.
mkdir -p gpurun_out/final
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_gowalla_steps20.json 2> gpurun_out/final/bench_gowalla_steps20.err
python bench.py > gpurun_out/final/bench_gowalla.json 2> gpurun_out/final/bench_gowalla.err
python bench.py --workload yelp2018 > gpurun_out/final/bench_lightgcn_yelp2018.json 2> gpurun_out/final/bench_lightgcn_yelp2018.err
python bench.py --workload ml10m --no-cpu-baseline > gpurun_out/final/bench_ml10m.json 2>/dev/null
python bench.py --workload addressa --no-cpu-baseline > gpurun_out/final/bench_addressa.json 2>/dev/null
python bench.py --train normalbce --no-cpu-baseline > gpurun_out/final/bench_gowalla_normalbce.json 2>/dev/null
python bench.py --no-defer --no-cpu-baseline > gpurun_out/final/bench_gowalla_nodefer.json 2>/dev/null
python bench.py --workload config4 --c4-lazy 1 --steps 40 --warmup 5 --regions 1 --no-eval --no-cpu-baseline > gpurun_out/final/bench_config4_1gpu_dense_adam.json 2>/dev/null
python bench.py --workload config4 --steps 256 --warmup 70 --regions 2 > gpurun_out/final/bench_config4_1gpu.json 2> gpurun_out/final/bench_config4_1gpu.err
MACR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --c4-users 2000000 --c4-items 200000 --c4-eval-users 20000 > gpurun_out/final/bench_gowalla_2ranks_gloo_one_gpu.json 2> gpurun_out/final/bench_gowalla_2ranks_gloo_one_gpu.err
tail -c 400 gpurun_out/final/bench_gowalla.json
