"""`from evaluator import eval_score_matrix_foldout` -- same entry point as the reference's
macr_lightgcn/evaluator/__init__.py:9 (Cython/C++), backed by the HIP kernels macr_topk_scores +
macr_metrics_foldout.  No build_ext step is needed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macr_amd.evaluator import eval_score_matrix_foldout  # noqa: E402,F401
