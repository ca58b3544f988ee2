#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "four_gib" 2>&1 | tail -12
