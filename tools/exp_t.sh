#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_product.py -x -q -k "trajectory or fullsize_train or sampler" 2>&1 | tail -15
