bash tools/profile.sh c4 python $PWD/bench.py --workload config4 --steps 128 --warmup 70 --regions 1 --no-cpu-baseline --no-eval
