timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "outside_the_window or large_logits" 2>&1 | tail -6
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
