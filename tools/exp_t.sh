#!/bin/bash
for k in 1 2 3 4 5 6; do
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "not lgcn and not lightgcn" 2>&1 | grep -E "passed|failed|Error|error|argument" | head -8
done
