#!/bin/bash
# All rocprofv3 evidence of a round in one gpurun call (kernel trace + PMC passes each, every pass under `timeout`):
#   bench (default workload), LightGCN at Yelp2018 shapes (SpMM), the pair kernels at B = 2^20.
bash tools/profile.sh bench python $PWD/bench.py --steps 200 --warmup 20 --regions 3 --no-cpu-baseline
bash tools/profile.sh lgcn python $PWD/bench.py --workload yelp2018 --steps 100 --warmup 10 --regions 3 --no-cpu-baseline --eval-reps 3
bash tools/profile.sh ml10m python $PWD/bench.py --workload ml10m --steps 100 --warmup 10 --regions 3 --no-cpu-baseline --no-e2e --eval-reps 3
bash tools/profile.sh c4 python $PWD/bench.py --workload config4 --steps 128 --warmup 70 --regions 1 --no-cpu-baseline --no-eval
bash tools/profile.sh pair20 python $PWD/tools/bench_pair_kernel.py 20
python tools/bench_pair_kernel.py > gpurun_out/pair_scale.json 2> gpurun_out/pair_scale.err
