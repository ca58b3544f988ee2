timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/ab_step.sh gpurun_out/r05_ab_neutral.txt "MACR_BXB_NEUTRAL=0" "-"
AB_ARGS="--workload ml10m" bash tools/ab_step.sh gpurun_out/r05_ab_neutral_ml10m.txt "MACR_BXB_NEUTRAL=0" "-"
AB_ARGS="--steps 200 --warmup 20 --regions 3 --eval-train-steps 0" bash tools/ab_step.sh gpurun_out/r05_ab_neutral_early.txt "MACR_BXB_NEUTRAL=0" "-"
