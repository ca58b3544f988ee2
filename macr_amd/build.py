"""Build libmacr_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmacr_hip.so")
COMPAT_LIB = os.path.join(CSRC, "libmacr_eval_compat.so")      # the reference's own evaluator ABI (include/macr_eval_compat.h)
IEEE_LIB = os.path.join(CSRC, "libmacr_hip_ieee.so")           # test rig: the Adam pass with IEEE sqrtf / division
TEST_LIB = os.path.join(CSRC, "libmacr_hip_test.so")           # test rig: + the entry points of include/macr_hip_test.h


def build(force=False, jobs=None):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(CSRC, "..", "..", "include", "macr_hip.h"))
    srcs.append(os.path.join(CSRC, "..", "..", "include", "macr_eval_compat.h"))
    srcs.append(os.path.join(CSRC, "..", "..", "include", "macr_hip_test.h"))
    srcs.append(os.path.join(CSRC, "Makefile"))
    newest = max(os.path.getmtime(s) for s in srcs)
    stale = any(not os.path.exists(l) or os.path.getmtime(l) < newest for l in (LIB, COMPAT_LIB, IEEE_LIB, TEST_LIB))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "-j%d" % (jobs or min(8, os.cpu_count() or 1))])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
