"""ctypes binding of libmacr_hip.so (the C ABI of include/macr_hip.h).

Loading fails LOUDLY when the extension has not been built: there is no
fallback implementation of the kernels anywhere in this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MACR_HIP_LIB lets kernel-development tools load an alternative build of the same ABI (ablations).
LIB_PATH = os.environ.get("MACR_HIP_LIB") or os.path.join(_HERE, "csrc", "libmacr_hip.so")

OK, E_INVALID, E_UNSUPPORTED, E_WORKSPACE, E_LAUNCH = 0, -1, -2, -3, -4
LOSS_NORMALBCE, LOSS_RUBIBCEBOTH, LOSS_RUBIBCE = 0, 1, 2
STEP_DEFER, STEP_PENDING, STEP_LOSS_ONLY, STEP_DENSE_LAYERS = 1, 2, 4, 8
SCORE_NORMAL, SCORE_RUBI_BOTH, SCORE_RUBI, SCORE_DIRECT_MINUS, SCORE_DIRECT_MINUS_BOTH = 0, 1, 2, 3, 4
MAX_TOPK = 128
MAX_TOPK_FUSED = 32
MAX_SWEEP = 4
ABI_VERSION = 15
LAZY_STATE_BYTES, LAZY_MAX_PERIOD = 1040, 64


class MacrError(RuntimeError):
    def __init__(self, code, msg):
        super(MacrError, self).__init__("macr_hip error %d: %s" % (code, msg))
        self.code = code


class Hyper(ctypes.Structure):
    """struct macr_hyper"""
    _fields_ = [("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("adam_eps", ctypes.c_float), ("decay", ctypes.c_float), ("alpha", ctypes.c_float),
                ("beta", ctypes.c_float), ("batch_size_cfg", ctypes.c_int32)]


class LazyAdam(ctypes.Structure):
    """struct macr_lazy_adam"""
    _fields_ = [("state", ctypes.c_void_p), ("stampP", ctypes.c_void_p), ("stampQ", ctypes.c_void_p), ("period", ctypes.c_int)]


_p, _i, _f, _z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_ll = ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/macr_hip.h declaration by declaration
SIGNATURES = {
    "macr_abi_version": (_i, []),
    "macr_last_error": (ctypes.c_char_p, []),
    "macr_build_info": (ctypes.c_char_p, []),
    "macr_timing_begin": (_i, [_p]),
    "macr_timing_end": (_i, [_i, _p, _p]),
    "macr_mf_train_workspace_bytes": (_z, [_i, _i]),
    "macr_mf_train_step": (_i, [_i] * 5 + [_p] * 3 + [_p] * 12 + [_p] * 4 + [_p, ctypes.POINTER(Hyper), _p, _i, _p, _z, _p]),
    "macr_mf_train_flush": (_i, [_i] * 5 + [_p] * 12 + [_p] * 4 + [ctypes.POINTER(Hyper), _p, _z, _p]),
    "macr_mf_train_step_lazy": (_i, [_i] * 5 + [_p] * 3 + [_p] * 12 + [_p] * 4 + [_p, ctypes.POINTER(Hyper), _p, _i, ctypes.POINTER(LazyAdam), _p, _z, _p]),
    "macr_mf_train_flush_lazy": (_i, [_i] * 5 + [_p] * 12 + [_p] * 4 + [ctypes.POINTER(Hyper), ctypes.POINTER(LazyAdam), _p, _z, _p]),
    "macr_shard_workspace_bytes": (_z, [_i, _i]),
    "macr_shard_gather": (_i, [_i, _i, _p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "macr_shard_forward": (_i, [_i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "macr_shard_bxb": (_i, [_i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "macr_shard_backward": (_i, [_i, _i, _i, _p, _p, _p, _p, ctypes.POINTER(Hyper), _p, _p, _p, _p, _z, _p]),
    "macr_shard_slice": (_i, [_i, _i, _i, _i, _p, _p]),
    "macr_shard_route": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "macr_shard_forward_slice": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "macr_shard_backward_slice": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, ctypes.POINTER(Hyper), _p, _p, _p, _p, _p, _z, _p]),
    "macr_shard_stage": (_i, [_i, _i, _p, _p, _z]),
    "macr_shard_apply": (_i, [_i] * 9 + [_p] * 3 + [_p] * 12 + [_p] * 4 + [ctypes.POINTER(Hyper), _p, _z, _p]),
    "macr_shard_gather_lazy": (_i, [_i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, ctypes.POINTER(Hyper),
                                    ctypes.POINTER(LazyAdam), _p, _p]),
    "macr_lazy_rows": (_i, [_ll, _i, _p, _p, _p, _p, _p, _p, ctypes.POINTER(Hyper), _p, _p]),
    "macr_shard_apply_lazy": (_i, [_i] * 9 + [_p] * 3 + [_p] * 12 + [_p] * 4 + [ctypes.POINTER(Hyper), ctypes.POINTER(LazyAdam), _p, _z, _p]),
    "macr_lazy_flush": (_i, [_i, _ll, _ll] + [_p] * 6 + [ctypes.POINTER(Hyper), ctypes.POINTER(LazyAdam), _p]),
    "macr_sample_triples": (_i, [ctypes.c_uint64, ctypes.c_uint64, _i, _i, _p, _i, _p, _p, _p, _p]),
    "macr_sample_triples_many": (_i, [ctypes.c_uint64, ctypes.c_uint64, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "macr_spmm_plan_bytes": (_z, [_i, _p, _p, _p]),
    "macr_spmm_plan_build": (_i, [_i, _p, _p, _p, _p, _z]),
    "macr_lgcn_work_floats": (_z, [_i, _i, _p]),
    "macr_lgcn_propagate": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "macr_lgcn_train_workspace_bytes": (_z, [_i, _i, _i, _p]),
    "macr_lgcn_train_step": (_i, [_i] * 6 + [_p] * 3 + [_p] * 2 + [_p] * 3 + [_p] * 9 + [_p, ctypes.POINTER(Hyper), _p, _i, _p, _z, _p]),
    "macr_lgcn_train_workspace_bytes_t": (_z, [_i, _i, _i, _p, _p]),
    "macr_lgcn_train_step_t": (_i, [_i] * 6 + [_p] * 3 + [_p] * 2 + [_p] * 3 + [_p] * 2 + [_p] * 3 + [_p] * 9 + [_p, ctypes.POINTER(Hyper), _p, _i, _p, _z, _p]),
    "macr_branch_sigmoid": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "macr_branch_sigmoid2": (_i, [_i, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p]),
    "macr_score_topk_splits": (_i, [_i, _i, _i]),
    "macr_score_topk_uses_seeds": (_i, [_i, _i, _i]),
    "macr_score_topk_workspace_bytes": (_z, [_i, _i, _i]),
    "macr_score_topk": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "macr_score_topk_first_round": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "macr_score_topk_repair_round": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "macr_score_topk_prologue": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "macr_score_topk_prologue_prep": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _z, _p]),
    "macr_score_topk_sweep_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "macr_score_topk_sweep": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _z, _p]),
    "macr_mask_bits_bytes": (_z, [_i, _i]),
    "macr_mask_bits_build": (_i, [_i, _i, _p, _p, _i, _p, _p]),
    "macr_score_matrix": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p]),
    "macr_topk_scores_workspace_bytes": (_z, [_i, _i]),
    "macr_topk_scores": (_i, [_p, _i, _i, _i, _p, _p, _p, _z, _p]),
    "macr_topk_merge": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "macr_metrics_foldout": (_i, [_i, _i, _p, _p, _p, _p, _i, _p]),
    "macr_metrics_foldout_fill": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _i, _p]),
    "macr_metrics_mf": (_i, [_i, _i, _p, _p, _p, _p, ctypes.POINTER(ctypes.c_int32), _i, _p, _p]),
    "macr_metrics_mf_mean_workspace_bytes": (_z, [_i, _i]),
    "macr_metrics_mf_mean": (_i, [_i, _i, _p, _p, _p, _p, ctypes.POINTER(ctypes.c_int32), _i, _p, _p, _p, _z, _p]),
    "macr_colmean": (_i, [_p, _i, _i, _i, _p, _p]),
}

# include/macr_hip_test.h: entry points of libmacr_hip_test.so only (tests/); the product library exports none of them
TEST_LIB_PATH = os.path.join(_HERE, "csrc", "libmacr_hip_test.so")
TEST_SIGNATURES = {
    "macr_test_bf16_products_workspace_bytes": (_z, [_i, _i, _i]),
    "macr_test_bf16_products": (_i, [_i, _i, _i, _p, _p, _f, _p, _p, _p, _z, _p]),
    "macr_test_bf16_scores_workspace_bytes": (_z, [_i, _i, _i]),
    "macr_test_bf16_scores": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _f, _p, _p, _p, _z, _p]),
    "macr_test_f16_scores": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _f, _p, _p, _p, _z, _p]),
}

_lib = None
_test = None


def test_lib():
    """libmacr_hip_test.so: the product sources + the test-only entry points (tests/ only)."""
    global _test
    if _test is None:
        _load_hip_runtime_first()
        if not os.path.exists(TEST_LIB_PATH):
            raise RuntimeError("macr_amd: %s is missing; build it with `python -m macr_amd.build`" % TEST_LIB_PATH)
        L = ctypes.CDLL(TEST_LIB_PATH)
        for name, (res, args) in TEST_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        L.macr_last_error.restype = ctypes.c_char_p
        _test = L
    return _test


def _load_hip_runtime_first():
    """PyTorch-ROCm ships its own libamdhip64 and must be the one that brings the HIP runtime into the process: if this
    library is loaded first it pulls in /opt/rocm's copy, torch then loads its bundled one, and kernels launched from
    here fail with "no ROCm-capable device is detected" (two runtimes in one process; seen when build() and smoke()
    run in the same interpreter).  Importing torch first makes the dynamic loader reuse the runtime torch loaded."""
    import torch  # noqa: F401


def lib():
    global _lib
    if _lib is None:
        _load_hip_runtime_first()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "macr_amd: HIP extension %s is missing.  Build it with "
                "`python -m macr_amd.build` (hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = L.macr_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("macr_amd: libmacr_hip.so ABI %d != expected %d; rebuild" % (got, ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise MacrError(rc, lib().macr_last_error().decode("utf-8", "replace"))


# ---- the reference's own evaluator ABI (include/macr_eval_compat.h): same names and signatures as
# macr_lightgcn/evaluator/cpp/include/tools.h:24 and evaluate_foldout.h:115-118
COMPAT_LIB_PATH = os.path.join(_HERE, "csrc", "libmacr_eval_compat.so")
COMPAT_SIGNATURES = {
    "c_top_k_array_index": (None, [_p, _i, _i, _i, _i, _p]),
    "evaluate_foldout": (None, [_i, _p, _i, _p, _p, _i, _p]),
    "macr_eval_compat_status": (_i, []),
    "macr_eval_compat_error": (ctypes.c_char_p, []),
}
_compat = None


def compat_lib():
    global _compat
    if _compat is None:
        _load_hip_runtime_first()
        if not os.path.exists(COMPAT_LIB_PATH):
            raise RuntimeError("macr_amd: %s is missing; build it with `python -m macr_amd.build`" % COMPAT_LIB_PATH)
        L = ctypes.CDLL(COMPAT_LIB_PATH)
        for name, (res, args) in COMPAT_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _compat = L
    return _compat
