"""The reference cythonizes its C++ evaluator here (`python setup.py build_ext --inplace`,
macr_lightgcn/setup.py:1-24).  The MI355X build has no Cython step: this script compiles the HIP
extension in-tree instead, so the documented command keeps working."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macr_amd.build import build  # noqa: E402

if __name__ == "__main__":
    print(build())
