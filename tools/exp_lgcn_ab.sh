#!/bin/bash
# A/B of the LightGCN step: row kernel / stream kernel, with and without the XCD-affine piece order (3 runs each)
for cfg in "row+octants:MACR_SPMM_STREAM=0" "stream+octants:MACR_SPMM_STREAM=1" "row:MACR_SPMM_STREAM=0 MACR_SPMM_OCTANTS=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for k in 1 2 3; do echo -n "$name "; env $envs python tools/bench_lgcn.py | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['us_per_step'],1), j['kernels_us'])"; done
done
