"""CPU-side checks of the C-ABI library: it builds, loads, and exports every
symbol include/macr_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from macr_amd import _lib
from macr_amd.build import build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build()
    return _lib.lib()


def header_symbols():
    src = open(os.path.join(REPO, "include", "macr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(macr_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = header_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SIGNATURES) == names          # the ctypes table mirrors the header exactly


def test_test_only_entry_points_live_in_their_own_library(lib):
    """include/macr_hip_test.h: exported by libmacr_hip_test.so (what tests load), by the product library not at all"""
    src = open(os.path.join(REPO, "include", "macr_hip_test.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = sorted(set(re.findall(r"\b(macr_[a-z0-9_]+)\s*\(", src)))
    assert declared == sorted(_lib.TEST_SIGNATURES) and len(declared) == 5
    T = _lib.test_lib()
    for name in declared:
        assert hasattr(T, name), name
        assert not hasattr(lib, name), "%s must not ship in libmacr_hip.so" % name
    assert not any(n.startswith("macr_test_") for n in header_symbols())
    assert T.macr_abi_version() == _lib.ABI_VERSION          # the same sources, the same ABI


def test_ranking_calls_take_the_filter_per_call(lib):
    """abi 10: no process-wide filter switch; an out-of-range filter is refused before any device work"""
    assert not hasattr(lib, "macr_set_eval_filter")
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.macr_score_topk(_lib.SCORE_NORMAL, 7, 4, 4, 64, p, None, p, None, None, 0.0, None, None, None, None, 0, 2, 1, None, None,
                             p, p, None, p, 1 << 20, None)
    assert rc == _lib.E_INVALID and b"filter=7" in lib.macr_last_error(), lib.macr_last_error()


def test_version_and_error_plumbing(lib):
    assert lib.macr_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.macr_build_info()
    # argument validation happens before any device work, so it is checkable without a GPU
    rc = lib.macr_topk_scores(None, 10, 10, 20, None, None, None, 0, None)
    assert rc == _lib.E_INVALID and b"null pointer" in lib.macr_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.macr_topk_scores(p, 10, 10, 129, p, None, None, 0, None)       # any K, but past 128 it wants its scratch
    assert rc == _lib.E_WORKSPACE and lib.macr_topk_scores_workspace_bytes(10, 129) >= 80
    assert lib.macr_topk_scores_workspace_bytes(10, 128) == 0
    rc = lib.macr_branch_sigmoid(p, None, 1, 48, p, p, None)
    assert rc == _lib.E_UNSUPPORTED and b"d=48" in lib.macr_last_error()
    assert lib.macr_mf_train_workspace_bytes(4096, 64) > 0
    assert lib.macr_mf_train_workspace_bytes(4096, 48) == 0
    assert 1 <= lib.macr_score_topk_splits(15424, 40981, 64) <= 64


def test_hyper_struct_layout():
    assert ctypes.sizeof(_lib.Hyper) == 32


def test_ops_reject_cpu_tensors():
    import torch
    from macr_amd import ops
    with pytest.raises(ops.MacrError):
        ops.branch_sigmoid(torch.zeros(4, 64), torch.zeros(64))


def test_reference_evaluator_abi_is_exported_with_its_own_names():
    """include/macr_eval_compat.h: the two functions the reference's .pyx binds (apt_evaluate_foldout.pyx:11-19)."""
    build()
    L = _lib.compat_lib()
    src = open(os.path.join(REPO, "include", "macr_eval_compat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = sorted(set(re.findall(r"\b([a-z_][a-z0-9_]+)\s*\(", src)) - {"defined"})
    assert declared == sorted(_lib.COMPAT_SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name
    # argument validation is checkable without a GPU: a failed call reports through the status channel
    L.c_top_k_array_index(None, 5, 2, 3, 1, None)
    assert L.macr_eval_compat_status() == _lib.E_INVALID and b"bad argument" in L.macr_eval_compat_error()


def test_filter_constants_agree_across_header_bindings_evaluator_and_bench():
    """MACR_EVAL_FILTER_* (include/macr_hip.h) = macr_amd/ops.py's table; the Evaluator's default filter is the one bench.py times
    as `eval_users_per_s`; the readiness flags are the header's."""
    import importlib.util
    from macr_amd import ops
    src = open(os.path.join(REPO, "include", "macr_hip.h")).read()
    consts = dict(re.findall(r"#define (MACR_EVAL_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)", src))
    assert int(consts["MACR_EVAL_FILTER_F32"], 0) == ops.EVAL_FILTER_F32 and int(consts["MACR_EVAL_FILTER_BF16"], 0) == ops.EVAL_FILTER_BF16
    assert int(consts["MACR_EVAL_FILTER_F16"], 0) == ops.EVAL_FILTER_F16 == ops.eval_filter_code("f16")
    assert int(consts["MACR_EVAL_WS_READY"], 0) == ops.EVAL_WS_READY and int(consts["MACR_EVAL_PREP_READY"], 0) == ops.EVAL_PREP_READY
    with pytest.raises(Exception):
        ops.eval_filter_code("fp16")                  # a typo is refused by name
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    ev_src = open(os.path.join(REPO, "macr_amd", "evaluator.py")).read()
    default = re.search(r'os\.environ\.get\("MACR_EVAL_FILTER", "([a-z0-9]+)"\)', ev_src).group(1)
    bench_src = open(spec.origin).read()
    assert re.search(r'DEFAULT_EVAL_FILTER = "%s"' % default, bench_src), default
