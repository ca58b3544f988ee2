timeout 900 python -m pytest tests/test_gpu_lazy_adam.py -x -q 2>&1 | tail -30
