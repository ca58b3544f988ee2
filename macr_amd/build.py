"""Build libmacr_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmacr_hip.so")


def build(force=False, jobs=None):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(CSRC, "..", "..", "include", "macr_hip.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(["make", "-s", "-C", CSRC, "-j%d" % (jobs or min(8, os.cpu_count() or 1))])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
