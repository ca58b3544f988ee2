timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pinned.py tests/test_gpu_lazy_adam.py -x -q -k "not topk and not rank" 2>&1 | tail -4
bash tools/ab_step.sh gpurun_out/r05_ab_partition.txt "MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_NOPARTITION.so" "-"
AB_ARGS="--workload ml10m" bash tools/ab_step.sh gpurun_out/r05_ab_partition_ml10m.txt "MACR_HIP_LIB=$PWD/macr_amd/csrc/_abl/libmacr_NOPARTITION.so" "-"
