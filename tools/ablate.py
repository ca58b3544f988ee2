#!/usr/bin/env python3
"""Kernel-development helper: build variants of libmacr_hip.so with -D switches and time them.

  python tools/ablate.py build NAME -DMACR_ABL_X ...     -> macr_amd/csrc/_abl/libmacr_hip_NAME.so
  MACR_HIP_LIB=<that .so> python bench.py ...            -> run any entry point against the variant
Variants are timing probes only (they may compute wrong results)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "macr_amd", "csrc")
SRCS = ["capi_common.hip", "train_kernels.hip", "spmm_kernels.hip", "eval_kernels.hip", "sample_kernels.hip"]


def build(name, defs):
    out = os.path.join(CSRC, "_abl")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libmacr_hip_%s.so" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
           "-shared", "-o", lib] + defs + [os.path.join(CSRC, s) for s in SRCS]
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if sys.argv[1] == "build":
        print(build(sys.argv[2], sys.argv[3:]))
