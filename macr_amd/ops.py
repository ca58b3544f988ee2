"""Torch-tensor face of the C ABI (include/macr_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every operator
below hands raw device pointers to libmacr_hip.so.  Nothing in this module has
a CPU or eager-PyTorch fallback -- CPU tensors are rejected.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (LOSS_NORMALBCE, LOSS_RUBIBCEBOTH, LOSS_RUBIBCE, SCORE_NORMAL, SCORE_RUBI_BOTH, SCORE_RUBI,  # noqa: F401
                   SCORE_DIRECT_MINUS, SCORE_DIRECT_MINUS_BOTH, MAX_TOPK,
                   Hyper, MacrError, check)


def _stream(device_index=None):
    """the current HIP stream of the device as a void*.  torch.cuda.current_stream() costs ~8 us of Python per call --
    a quarter of a 36 us training step -- so the per-step callers pass their device index and take the raw handle."""
    if device_index is not None:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(device_index))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("tensor required")
    if not t.is_cuda:
        raise MacrError(_lib.E_INVALID, "macr_amd.ops needs device (HIP) tensors; got a CPU tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


_f32, _i32, _f64 = torch.float32, torch.int32, torch.float64


def _ptr_visible(t, dtype=None, allow_none=False):
    """A buffer a kernel WRITES a few words of results to: a device tensor, or a PINNED host tensor -- pinned host memory
    is mapped into the device's address space, the kernel's stores land in it directly and the host reads them after
    synchronising with the stream (no copy launched)."""
    if t is None or t.is_cuda:
        return _ptr(t, dtype, allow_none)
    if not t.is_pinned():
        raise MacrError(_lib.E_INVALID, "macr_amd.ops: a host result buffer must be pinned (tensor.pin_memory())")
    if dtype is not None and t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def _require_f32(**tensors):
    """the kernels read fp32 tables through raw pointers: any other dtype is refused here, not reinterpreted"""
    for name, t in tensors.items():
        if t.dtype != _f32:
            raise TypeError("%s must be float32, got %s" % (name, t.dtype))


class CSR(object):
    """int32 CSR on the device: ptr[rows+1], idx[nnz] (+ optional fp32 val[nnz])."""

    def __init__(self, ptr, idx, val=None):
        self.ptr, self.idx, self.val = ptr, idx, val
        self.plan_host = self.plan_dev = None          # SpMM plan (adjacency matrices only)

    def build_spmm_plan(self):
        """Static schedule of the SpMM kernels (macr_spmm_plan_build): the row items with the hub rows cut into pieces
        and, when the matrix has values, the entry stream of the dense layers; built once per graph."""
        import numpy as np
        rowptr = np.ascontiguousarray(self.ptr.cpu().numpy(), dtype=np.int32)
        N = len(rowptr) - 1
        col = val = None
        if self.val is not None:
            col = np.ascontiguousarray(self.idx.cpu().numpy(), dtype=np.int32)
            val = np.ascontiguousarray(self.val.cpu().numpy(), dtype=np.float32)
        hp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        L = _lib.lib()
        nbytes = L.macr_spmm_plan_bytes(N, hp(rowptr), hp(col), hp(val))
        host = np.zeros((nbytes + 3) // 4, np.int32)
        check(L.macr_spmm_plan_build(N, hp(rowptr), hp(col), hp(val), hp(host), host.nbytes))
        self.plan_host = host                            # keep alive: the launcher reads its headers
        self.plan_dev = torch.from_numpy(host).to(self.ptr.device)      # (torch allocations are 256-byte aligned)
        return self

    def mask_bits(self, U, n_local, item_offset):
        """(item tile, row) bitmap of this CSR used as a ranking mask (macr_mask_bits_build); cached: the train
        lists an evaluator masks never change."""
        key = (U, n_local, item_offset)
        cache = self.__dict__.setdefault("_mask_bits", {})
        bits = cache.get(key)
        if bits is None:
            if len(cache) >= 4:
                cache.clear()
            nbytes = _lib.lib().macr_mask_bits_bytes(U, n_local)
            bits = cache[key] = torch.empty(nbytes // 4, dtype=_i32, device=self.ptr.device)
            check(_lib.lib().macr_mask_bits_build(U, n_local, _ptr(self.ptr, _i32), _ptr(self.idx, _i32), item_offset,
                                                  _ptr(bits), _stream()))
        return bits

    def _plan_ptrs(self):
        if self.plan_host is None:
            return None, None
        return ctypes.c_void_p(self.plan_dev.data_ptr()), self.plan_host.ctypes.data_as(ctypes.c_void_p)

    @staticmethod
    def from_lists(lists, device, sort=True):
        ptr = [0]
        flat = []
        for row in lists:
            row = sorted(row) if sort else list(row)
            flat.extend(row)
            ptr.append(len(flat))
        idx = torch.tensor(flat if flat else [0], dtype=_i32, device=device)
        return CSR(torch.tensor(ptr, dtype=_i32, device=device), idx)

    @staticmethod
    def from_scipy(m, device):
        m = m.tocsr()
        m.sort_indices()
        return CSR(torch.from_numpy(m.indptr.astype("int32")).to(device),
                   torch.from_numpy(m.indices.astype("int32")).to(device),
                   torch.from_numpy(m.data.astype("float32")).to(device)).build_spmm_plan()

    def row_range(self, a, b):
        """CSR of the contiguous rows [a, b) (cached: evaluators rank the same query chunks every epoch)."""
        cache = self.__dict__.setdefault("_row_ranges", {})
        sub = cache.get((a, b))
        if sub is None:
            lo, hi = int(self.ptr[a]), int(self.ptr[b])
            idx = self.idx[lo:hi] if hi > lo else self.idx[:1]
            sub = cache[(a, b)] = CSR((self.ptr[a:b + 1] - lo).contiguous(), idx.contiguous())
        return sub

    def rows(self, sel):
        """sub-CSR for the given row selection (host-side index list/tensor)."""
        sel = torch.as_tensor(sel, dtype=torch.long, device=self.ptr.device)
        starts, ends = self.ptr[sel].long(), self.ptr[sel + 1].long()
        lens = ends - starts
        ptr = torch.zeros(len(sel) + 1, dtype=torch.long, device=self.ptr.device)
        ptr[1:] = torch.cumsum(lens, 0)
        total = int(ptr[-1])
        if total == 0:
            return CSR(ptr.to(_i32), torch.zeros(1, dtype=_i32, device=self.ptr.device))
        rowid = torch.repeat_interleave(torch.arange(len(sel), device=self.ptr.device), lens)
        pos = torch.arange(total, device=self.ptr.device) - ptr[rowid] + starts[rowid]
        return CSR(ptr.to(_i32), self.idx[pos].contiguous())


# ------------------------------------------------------------------ per-kernel timing (benchmarks)
def timing_begin():
    check(_lib.lib().macr_timing_begin(_stream()))


def timing_end(max_n=256):
    """-> list of (kernel name, milliseconds) for every launch since timing_begin (synchronises)."""
    names = ctypes.create_string_buffer(32 * max_n)
    ms = (ctypes.c_float * max_n)()
    n = _lib.lib().macr_timing_end(max_n, ctypes.cast(names, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p))
    raw = names.raw
    return [(raw[k * 32:(k + 1) * 32].split(b"\0", 1)[0].decode(), float(ms[k])) for k in range(n)]


# ------------------------------------------------------------------ evaluator ops
def branch_sigmoid(rows, w, idx=None):
    """sigmoid(rows[idx] . w)  (macr_mf/model.py:194-196,:199)."""
    n = rows.shape[0] if idx is None else idx.numel()
    out = torch.empty(n, dtype=_f32, device=rows.device)
    check(_lib.lib().macr_branch_sigmoid(_ptr(rows, _f32), _ptr(idx, _i32, True), n, rows.shape[1],
                                         _ptr(w.reshape(-1), _f32), _ptr(out), _stream()))
    return out


def branch_sigmoid2(rows_a, w_a, idx_a, rows_b, w_b, idx_b):
    """(sigmoid(rows_a[idx_a] . w_a), sigmoid(rows_b[idx_b] . w_b)) in one launch."""
    n_a = rows_a.shape[0] if idx_a is None else idx_a.numel()
    n_b = rows_b.shape[0] if idx_b is None else idx_b.numel()
    out_a = torch.empty(n_a, dtype=_f32, device=rows_a.device)
    out_b = torch.empty(n_b, dtype=_f32, device=rows_b.device)
    check(_lib.lib().macr_branch_sigmoid2(rows_a.shape[1], _ptr(rows_a, _f32), _ptr(idx_a, _i32, True), n_a, _ptr(w_a.reshape(-1), _f32), _ptr(out_a),
                                          _ptr(rows_b, _f32), _ptr(idx_b, _i32, True), n_b, _ptr(w_b.reshape(-1), _f32), _ptr(out_b), _stream()))
    return out_a, out_b


def score_topk_splits(U, n_local, d):
    return _lib.lib().macr_score_topk_splits(U, n_local, d)


EVAL_FILTER_ENV, EVAL_FILTER_F32, EVAL_FILTER_BF16, EVAL_FILTER_F16 = 0, 1, 2, 3


_FILTER_NAMES = {"env": EVAL_FILTER_ENV, "f32": EVAL_FILTER_F32, "bf16": EVAL_FILTER_BF16, "f16": EVAL_FILTER_F16}


def eval_filter_code(mode):
    """"env" | "f32" | "bf16" | "f16" (any case) or 0..3 -> MACR_EVAL_FILTER_*; anything else is refused by name (a typo in
    MACR_EVAL_FILTER must not surface as a bare ValueError inside the first evaluation)."""
    if isinstance(mode, str) and mode.strip().lower() in _FILTER_NAMES:
        return _FILTER_NAMES[mode.strip().lower()]
    if isinstance(mode, int) and not isinstance(mode, bool) and mode in _FILTER_NAMES.values():
        return mode
    raise MacrError(_lib.E_INVALID, "eval filter %r: accepted values are 'env', 'f32', 'bf16', 'f16' (or 0, 1, 2, 3)" % (mode,))


_default_filter = EVAL_FILTER_ENV


def set_eval_filter(mode):
    """The filter ranking calls of THIS MODULE pass when their caller names none (`filter=None`): "f32", "bf16", "f16" or "env"
    (MACR_EVAL_FILTER in the environment, f32 when unset).  Host-side convenience for tests and tools; the C ABI takes the
    filter per call (include/macr_hip.h, abi 10) and keeps no state."""
    global _default_filter
    _default_filter = eval_filter_code(mode)


def _filter_arg(filter):
    return _default_filter if filter is None else eval_filter_code(filter)


def _check_test(rc):
    if rc != _lib.OK:
        raise MacrError(rc, _lib.test_lib().macr_last_error().decode("utf-8", "replace"))


def test_bf16_products(users, items, c=0.0):
    """TEST-ONLY (macr_test_bf16_products): the raw product of the bf16 filter's kernels for every (user row, item row)
    pair and the margin the filter grants each user row -> ((U, N) fp32, (U,) fp32)."""
    _require_f32(users=users, items=items)
    U, d = users.shape
    N = items.shape[0]
    L = _lib.test_lib()
    ws = torch.empty(L.macr_test_bf16_products_workspace_bytes(d, U, N), dtype=torch.uint8, device=items.device)
    prod = torch.empty((U, N), dtype=_f32, device=items.device)
    margin = torch.empty(U, dtype=_f32, device=items.device)
    _check_test(L.macr_test_bf16_products(d, U, N, _ptr(users, _f32), _ptr(items, _f32), float(c), _ptr(prod), _ptr(margin),
                                    _ptr(ws), ws.numel(), _stream()))
    return prod, margin


def test_bf16_scores(kind, users, items, sig_u=None, sig_i=None, c=0.0):
    """TEST-ONLY (macr_test_bf16_scores): the score the bf16 listing pass of macr_score_topk would list for every
    (user row, item row) pair of `kind`, and the margin the filter grants each user row -> ((U, N) fp32, (U,) fp32)."""
    _require_f32(users=users, items=items)
    U, d = users.shape
    N = items.shape[0]
    L = _lib.test_lib()
    ws = torch.empty(L.macr_test_bf16_scores_workspace_bytes(d, U, N), dtype=torch.uint8, device=items.device)
    out = torch.empty((U, N), dtype=_f32, device=items.device)
    margin = torch.empty(U, dtype=_f32, device=items.device)
    _check_test(L.macr_test_bf16_scores(kind, d, U, N, _ptr(users, _f32), _ptr(items, _f32), _ptr(sig_u, _f32, True), _ptr(sig_i, _f32, True),
                                  float(c), _ptr(out), _ptr(margin), _ptr(ws), ws.numel(), _stream()))
    return out, margin


def test_f16_scores(kind, users, items, sig_u=None, sig_i=None, c=0.0):
    """TEST-ONLY (macr_test_f16_scores): the same for the fp16 filter's listing pass (k_score_stream_h's MFMA sequence on
    the fp16 operand copies) -> ((U, N) fp32, (U,) fp32); a query whose scaled row leaves fp16's range has margin +inf."""
    _require_f32(users=users, items=items)
    U, d = users.shape
    N = items.shape[0]
    L = _lib.test_lib()
    ws = torch.empty(L.macr_test_bf16_scores_workspace_bytes(d, U, N), dtype=torch.uint8, device=items.device)
    out = torch.empty((U, N), dtype=_f32, device=items.device)
    margin = torch.empty(U, dtype=_f32, device=items.device)
    _check_test(L.macr_test_f16_scores(kind, d, U, N, _ptr(users, _f32), _ptr(items, _f32), _ptr(sig_u, _f32, True), _ptr(sig_i, _f32, True),
                                 float(c), _ptr(out), _ptr(margin), _ptr(ws), ws.numel(), _stream()))
    return out, margin


SEED_WIDTH = 32          # MACR_SEED_WIDTH
_topk_ws_cache = {}


def _topk_workspace(U, n_local, d, device):
    """Scratch of macr_score_topk (candidate lists); one buffer per device, grown on demand."""
    need = _lib.lib().macr_score_topk_workspace_bytes(U, n_local, d)
    ws = _topk_ws_cache.get(device)
    if ws is None or ws.numel() < need:
        ws = _topk_ws_cache[device] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


def _c_args(c):
    """c as (float by value, device pointer or None): a 1-element fp32 device tensor is read by the kernels at run time."""
    if isinstance(c, torch.Tensor):
        return 0.0, _ptr(c, _f32)
    return float(c), None


EVAL_WS_READY = 0x100
EVAL_PREP_READY = 0x200


def score_topk_prologue(users_tab, user_ids, items, K, w_item, w_user=None, seeded_first_round=False, filter=None):
    """The launches an evaluation of the two-branch scores starts with, as one (macr_score_topk_prologue): returns
    (sig_i, sig_u) -- branch_sigmoid(items, w_item), branch_sigmoid(users_tab, w_user, user_ids) or None -- and leaves the head
    of the device's ranking workspace initialised for the score_topk call that follows with ws_ready=True and the same U,
    items, K, filter (seeded_first_round: that call is first_round=True with seeds)."""
    U = users_tab.shape[0] if user_ids is None else user_ids.numel()
    n_local, d = items.shape
    sig_i = torch.empty(n_local, dtype=_f32, device=items.device)
    sig_u = torch.empty(U, dtype=_f32, device=items.device) if w_user is not None else None
    ws = _topk_workspace(U, n_local, d, items.device)
    check(_lib.lib().macr_score_topk_prologue(_filter_arg(filter), U, n_local, d, K, int(bool(seeded_first_round)),
                                              _ptr(items, _f32), _ptr(w_item.reshape(-1), _f32), _ptr(sig_i),
                                              _ptr(users_tab, _f32) if w_user is not None else None,
                                              _ptr(user_ids, _i32, True) if w_user is not None else None,
                                              _ptr(w_user.reshape(-1), _f32) if w_user is not None else None, _ptr(sig_u, _f32, True),
                                              _ptr(ws), ws.numel(), _stream()))
    return sig_i, sig_u


def score_topk_prologue_prep(kind, users_tab, user_ids, items, K, w_item, w_user=None, c=0.0, seeded_first_round=False):
    """score_topk_prologue AND the fp16 filter's operand copies in one launch (macr_score_topk_prologue_prep): every row is
    read once.  For the score_topk call that follows with filter="f16", ws_ready=True, prep_ready=True and the same kind, c,
    tables, K.  c: a float or a 1-element device tensor."""
    U = users_tab.shape[0] if user_ids is None else user_ids.numel()
    n_local, d = items.shape
    sig_i = torch.empty(n_local, dtype=_f32, device=items.device)
    sig_u = torch.empty(U, dtype=_f32, device=items.device) if w_user is not None else None
    ws = _topk_workspace(U, n_local, d, items.device)
    cv, cp = _c_args(c)
    check(_lib.lib().macr_score_topk_prologue_prep(kind, U, n_local, d, K, int(bool(seeded_first_round)),
                                                   _ptr(items, _f32), _ptr(w_item.reshape(-1), _f32), _ptr(sig_i),
                                                   _ptr(users_tab, _f32), _ptr(user_ids, _i32, True),
                                                   _ptr(w_user.reshape(-1), _f32) if w_user is not None else None, _ptr(sig_u, _f32, True),
                                                   cv, cp, _ptr(ws), ws.numel(), _stream()))
    return sig_i, sig_u


def score_topk(kind, users_tab, user_ids, items, K, sig_u=None, sig_i=None, c=0.0, mask=None,
               item_offset=0, n_splits=0, seed=None, seed_out=None, stats=None, first_round=False, repair_of=None,
               filter=None, ws_ready=False, prep_ready=False):
    """Fused U.I^T + epilogue + mask + top-K.  Returns (vals, idx) of shape (n_splits, U, K).
    c: python float, or a 1-element fp32 device tensor (read at run time: graph replays follow its value).
    seed: optional (U, SEED_WIDTH) int32 device tensor of global item ids per query -- what seed_out received last time:
    thresholds then come from the seeds' exact scores instead of a sampling pass -- same result, less time.
    seed_out: optional (U, SEED_WIDTH) int32 device tensor <- the best candidates per query (may be `seed` itself).
    filter: "f32" | "bf16" | "env" candidate filter of the listing pass (None: this module's default, set_eval_filter).
    stats: optional int32[2] device (or pinned host) tensor <- (query blocks listed twice because a threshold was too
    loose, 1 if the exact fallback kernel ran).
    first_round: macr_score_topk_first_round -- the first round alone, without the launches of the repair round and the
    fallback; stats (required) then says whether the result stands: stats[0] == 0, else finish it with
    repair_of=(vals, idx, ws) -- macr_score_topk_repair_round on the same arguments, the first-round call's outputs AND the
    workspace it ran on (thresholds, overflow counters and candidate lists are where the first round left them; the
    per-device cache may have been regrown by another caller since) -- or run the complete call.
    ws_ready: score_topk_prologue ran for this call (same stream, nothing of the ranking workspace touched in between);
    prep_ready: it was score_topk_prologue_prep (filter "f16": the operand copies are in the workspace too)."""
    U = users_tab.shape[0] if user_ids is None else user_ids.numel()
    n_local, d = items.shape
    if n_splits <= 0:
        n_splits = score_topk_splits(U, n_local, d)
    ws = None
    if repair_of is not None:
        vals, idx, ws = repair_of
        assert vals.shape == (n_splits, U, K) and idx.shape == (n_splits, U, K)
        assert ws is not None and ws.numel() >= _lib.lib().macr_score_topk_workspace_bytes(U, n_local, d)
    else:
        vals = torch.empty((n_splits, U, K), dtype=_f32, device=items.device)
        idx = torch.empty((n_splits, U, K), dtype=_i32, device=items.device)
    mp = _ptr(mask.ptr, _i32) if mask is not None else None
    mi = _ptr(mask.idx, _i32) if mask is not None else None
    mb = _ptr(mask.mask_bits(U, n_local, item_offset)) if mask is not None else None
    if ws is None:
        ws = _topk_workspace(U, n_local, d, items.device)
    cv, cp = _c_args(c)
    fn = _lib.lib().macr_score_topk_first_round if first_round else _lib.lib().macr_score_topk
    if repair_of is not None:
        fn = _lib.lib().macr_score_topk_repair_round
    assert not (ws_ready and repair_of is not None)
    assert not prep_ready or ws_ready
    check(fn(kind, _filter_arg(filter) | (EVAL_WS_READY if ws_ready else 0) | (EVAL_PREP_READY if prep_ready else 0), U, n_local, d,
             _ptr(users_tab, _f32), _ptr(user_ids, _i32, True),
             _ptr(items, _f32), _ptr(sig_u, _f32, True), _ptr(sig_i, _f32, True),
             cv, cp, mp, mi, mb, item_offset, K, n_splits, _ptr(seed, _i32, True), _ptr(seed_out, _i32, True), _ptr(vals), _ptr(idx),
             _ptr_visible(stats, _i32, True), _ptr(ws), ws.numel(), _stream()))
    return vals, idx


_sweep_ws_cache = {}


def score_topk_sweep(kind, users_tab, user_ids, items, K, sig_u, sig_i, c_dev, mask=None, item_offset=0, filter=None):
    """The fused ranking for SEVERAL values of c with ONE listing pass (macr_score_topk_sweep; tune.py:545-578).
    c_dev: fp32 device tensor of 1.._lib.MAX_SWEEP values.  Returns (vals, idx) of shape (n_c, U, K)."""
    U = users_tab.shape[0] if user_ids is None else user_ids.numel()
    n_local, d = items.shape
    n_c = c_dev.numel()
    vals = torch.empty((n_c, U, K), dtype=_f32, device=items.device)
    idx = torch.empty((n_c, U, K), dtype=_i32, device=items.device)
    mp = _ptr(mask.ptr, _i32) if mask is not None else None
    mi = _ptr(mask.idx, _i32) if mask is not None else None
    mb = _ptr(mask.mask_bits(U, n_local, item_offset)) if mask is not None else None
    need = _lib.lib().macr_score_topk_sweep_workspace_bytes(U, n_local, d, n_c)
    ws = _sweep_ws_cache.get(items.device)
    if ws is None or ws.numel() < need:
        ws = _sweep_ws_cache[items.device] = torch.empty(need, dtype=torch.uint8, device=items.device)
    check(_lib.lib().macr_score_topk_sweep(kind, _filter_arg(filter), U, n_local, d, _ptr(users_tab, _f32), _ptr(user_ids, _i32, True),
                                           _ptr(items, _f32), _ptr(sig_u, _f32, True), _ptr(sig_i, _f32), n_c,
                                           _ptr(c_dev, _f32), mp, mi, mb, item_offset, K, _ptr(vals), _ptr(idx),
                                           _ptr(ws), ws.numel(), _stream()))
    return vals, idx


def score_matrix(kind, users_tab, user_ids, items, sig_u=None, sig_i=None, c=0.0):
    U = users_tab.shape[0] if user_ids is None else user_ids.numel()
    n_local, d = items.shape
    out = torch.empty((U, n_local), dtype=_f32, device=items.device)
    cv, cp = _c_args(c)
    check(_lib.lib().macr_score_matrix(kind, U, n_local, d, _ptr(users_tab, _f32), _ptr(user_ids, _i32, True),
                                       _ptr(items, _f32), _ptr(sig_u, _f32, True), _ptr(sig_i, _f32, True),
                                       cv, cp, _ptr(out), _stream()))
    return out


def topk_scores(scores, K, want_vals=True):
    """Top-K column ids of every row (replaces c_top_k_array_index, tools.h:24); any K >= 1, as the reference's."""
    rows, cols = scores.shape
    if K < 1:
        raise MacrError(_lib.E_INVALID, "topk_scores: K=%d" % K)
    idx = torch.empty((rows, K), dtype=_i32, device=scores.device)
    val = torch.empty((rows, K), dtype=_f32, device=scores.device) if want_vals else None
    need = _lib.lib().macr_topk_scores_workspace_bytes(rows, K)
    ws = torch.empty(need, dtype=torch.uint8, device=scores.device) if need else None
    check(_lib.lib().macr_topk_scores(_ptr(scores, _f32), cols, rows, K, _ptr(idx), _ptr(val, None, True),
                                      _ptr(ws, None, True), need, _stream()))
    return idx, val


def topk_merge(vals, idxs, fill_mask=None):
    """(W,U,K) lists -> (U,K) + count of real candidates."""
    W, U, K = vals.shape
    ov = torch.empty((U, K), dtype=_f32, device=vals.device)
    oi = torch.empty((U, K), dtype=_i32, device=vals.device)
    oc = torch.empty(U, dtype=_i32, device=vals.device)
    fp = _ptr(fill_mask.ptr, _i32) if fill_mask is not None else None
    fi = _ptr(fill_mask.idx, _i32) if fill_mask is not None else None
    check(_lib.lib().macr_topk_merge(W, U, K, _ptr(vals, _f32), _ptr(idxs, _i32), fp, fi, _ptr(ov), _ptr(oi),
                                     _ptr(oc), _stream()))
    return ov, oi, oc


def metrics_foldout(rankings, gt, hr_in_ap_slot=False, fill_mask=None):
    """(U,K) rankings + ground-truth CSR -> (U,5K) fp32 (replaces evaluate_foldout, evaluate_foldout.h:115).
    hr_in_ap_slot: also apply batch_test.py:143-149 (ap block := 1[recall@k != 0]).
    fill_mask: lists with fewer than K ids (-1 from their first unused slot on) are completed with the query's masked
    ids, ascending -- topk_merge's fill without its launch."""
    U, K = rankings.shape
    out = torch.empty((U, 5 * K), dtype=_f32, device=rankings.device)
    if fill_mask is not None:
        check(_lib.lib().macr_metrics_foldout_fill(U, K, _ptr(rankings, _i32), _ptr(fill_mask.ptr, _i32), _ptr(fill_mask.idx, _i32),
                                                   _ptr(gt.ptr, _i32), _ptr(gt.idx, _i32), _ptr(out), int(hr_in_ap_slot), _stream()))
        return out
    check(_lib.lib().macr_metrics_foldout(U, K, _ptr(rankings, _i32), _ptr(gt.ptr, _i32), _ptr(gt.idx, _i32),
                                          _ptr(out), int(hr_in_ap_slot), _stream()))
    return out


def metrics_mf(rankings, cnt, gt, Ks):
    """(U,Kmax) rankings -> (U,4,len(Ks)) float64 {precision, recall, ndcg, hit} (macr_mf/train.py:32-117).
    cnt None: a list's length is its number of ids >= 0 (-1 = unused slot)."""
    U, Kmax = rankings.shape
    ks = (ctypes.c_int32 * len(Ks))(*[int(k) for k in Ks])
    out = torch.empty((U, 4, len(Ks)), dtype=_f64, device=rankings.device)
    check(_lib.lib().macr_metrics_mf(U, Kmax, _ptr(rankings, _i32), _ptr(cnt, _i32, True), _ptr(gt.ptr, _i32),
                                     _ptr(gt.idx, _i32), ks, len(Ks), _ptr(out), _stream()))
    return out


def metrics_mf_mean_workspace(U, nK, device):
    """the zero-filled scratch of metrics_mf_mean (ticket + one partial row per block of 64 queries)"""
    return torch.zeros(_lib.lib().macr_metrics_mf_mean_workspace_bytes(U, nK), dtype=torch.uint8, device=device)


def metrics_mf_mean(rankings, cnt, gt, Ks, ws, out=None, per_user=False):
    """colmean(metrics_mf(...)) in ONE launch (macr_metrics_mf_mean): (4, len(Ks)) float64 means over the query users
    (macr_mf/train.py:286-290); with per_user also the (U,4,len(Ks)) values.  ws: metrics_mf_mean_workspace(U, len(Ks)) -- one
    call at a time per buffer; out: optional float64 device or pinned host tensor of 4*len(Ks) elements."""
    U, Kmax = rankings.shape
    ks = (ctypes.c_int32 * len(Ks))(*[int(k) for k in Ks])
    if out is None:
        out = torch.empty(4 * len(Ks), dtype=_f64, device=rankings.device)
    pu = torch.empty((U, 4, len(Ks)), dtype=_f64, device=rankings.device) if per_user else None
    check(_lib.lib().macr_metrics_mf_mean(U, Kmax, _ptr(rankings, _i32), _ptr(cnt, _i32, True), _ptr(gt.ptr, _i32),
                                          _ptr(gt.idx, _i32), ks, len(Ks), _ptr(pu, _f64, True), _ptr_visible(out, _f64),
                                          _ptr(ws), ws.numel(), _stream()))
    m = out.reshape(4, len(Ks))
    return (m, pu) if per_user else m


def colmean(x, out=None):
    """out: optional float64 device or pinned host tensor of x.shape[1:]"""
    rows = x.shape[0]
    cols = x.numel() // rows
    if out is None:
        out = torch.empty(cols, dtype=_f64, device=x.device)
    if x.dtype not in (_f32, _f64):
        raise TypeError("colmean: fp32 or fp64")
    check(_lib.lib().macr_colmean(_ptr(x), 1 if x.dtype == _f32 else 0, rows, cols, _ptr_visible(out, _f64), _stream()))
    return out.reshape(x.shape[1:])


# ------------------------------------------------------------------ embedding widths
SUPPORTED_DIMS = (32, 64, 128, 256)


def padded_dim(d):
    """The width the kernels run an `--embed_size d` model at: the next of 32 / 64 / 128 / 256.  The host models keep
    their tables (and branch vectors) at that width with the extra columns ZERO; they stay exactly zero through training
    (every term of their gradient is a product with a zero column, Adam of a zero gradient on zero slots is zero) and add
    exact zeros to every dot product, so losses, parameters and rankings are those of the d-wide model
    (the reference accepts any integer: macr_mf/parse.py:27, utility/parser.py:32)."""
    for s in SUPPORTED_DIMS:
        if d <= s:
            return s
    raise MacrError(_lib.E_UNSUPPORTED, "embed_size %d > %d" % (d, SUPPORTED_DIMS[-1]))


def pad_cols(t, width):
    """(rows, d) or (d,) tensor -> contiguous tensor of `width` columns, zero-filled on the right"""
    if t.shape[-1] == width:
        return t.contiguous()
    out = torch.zeros(t.shape[:-1] + (width,), dtype=t.dtype, device=t.device)
    out[..., :t.shape[-1]] = t
    return out


# ------------------------------------------------------------------ LightGCN propagation
def lgcn_propagate(adj, E0, n_layers, out=None, work=None):
    """E = mean(E0, A E0, ..., A^L E0)  (LightGCN.py:288-309); also its own backward."""
    N, d = E0.shape
    if out is None:
        out = torch.empty_like(E0)
    pd, ph = adj._plan_ptrs()
    need = _lib.lib().macr_lgcn_work_floats(N, d, ph)
    if work is None or work.numel() < need:
        work = torch.zeros(need, dtype=_f32, device=E0.device)       # zero once: the hub rows' arrival counters live in it
    check(_lib.lib().macr_lgcn_propagate(N, d, n_layers, _ptr(adj.ptr, _i32), _ptr(adj.idx, _i32),
                                         _ptr(adj.val, _f32), pd, ph, _ptr(E0, _f32), _ptr(out, _f32),
                                         _ptr(work, _f32), _stream()))
    return out


# ------------------------------------------------------------------ training state
def make_hyper(lr, decay, alpha, beta, batch_size_cfg, beta1=0.9, beta2=0.999, adam_eps=1e-8):
    return Hyper(lr, beta1, beta2, adam_eps, decay, alpha, beta, int(batch_size_cfg))


def mf_lazy_period_for(n_rows, d, B):
    """K of the lazy dense Adam pass for deferred MF training (1 = off).  The dense pass rides in the (B,B) launch: 24*d bytes
    per row at ~6.2 TB/s beside 2*B*B pair terms at ~2.2 T/s.  Lazily the launch moves the rows of two batches and a K-th of
    the tables -- but the arithmetic of the pass stays (every row still receives every step), the forward kernel pays for the
    steps its rows are behind, and the flagged rows of a chunk fill a quarter of their waves' lanes.  Measured on the Gowalla
    shape (17.5 us of traffic beside 15.2 us of arithmetic; profiles/r05_lazy_mf_ab.txt): 34.6 us per step dense, 37.3 / 38.9 /
    43.7 with K = 2 / 4 / 8 -- so the lazy form is for tables whose pass dwarfs the (B,B) term (millions of rows on one GPU),
    where it is the difference between a step bound by 24*d*rows bytes and one bound by its batch."""
    adam_us = 24.0 * d * n_rows / 6.2e6
    bxb_us = max(2.0 * B * B / 2.2e6, 20.0)
    if adam_us < 4.0 * bxb_us:
        return 1
    return int(min(64, max(2, round(adam_us / bxb_us / 2.0))))


class MFState(object):
    """Device state of the MF model + its optimizer (one tf.train.AdamOptimizer instance).

    P,Q,w,wu mirror weights['user_embedding'], ['item_embedding'], self.w, self.w_user of
    macr_mf/model.py:107-122,:59-60.  Adam slots start at zero, beta powers at (beta1,beta2)."""

    def __init__(self, P, Q, w, wu, hyper, batch_cap, lazy_period=None):
        """lazy_period: K of the lazy dense Adam pass in deferred mode (include/macr_hip.h: macr_lazy_adam; 1 = the dense pass
        every step); None: the environment's MACR_LAZY_ADAM_MF, else by table and batch size (mf_lazy_period_for)"""
        _require_f32(P=P, Q=Q, w=w, wu=wu)
        dev = P.device
        if lazy_period is None and os.environ.get("MACR_LAZY_ADAM_MF", ""):
            lazy_period = int(os.environ["MACR_LAZY_ADAM_MF"])
        self.lazy_period = lazy_period        # None: decided per deferred sequence from its batch size
        self._seq_lazy = None                 # struct macr_lazy_adam of the running deferred sequence (None: the dense form)
        self._lazy_bufs = None
        self._P, self._Q = P.contiguous(), Q.contiguous()
        self.w, self.wu = w.reshape(-1).contiguous(), wu.reshape(-1).contiguous()
        self.hyper = hyper
        self.d = P.shape[1]
        z = torch.zeros_like
        self.mP, self.vP, self.mQ, self.vQ = z(self._P), z(self._P), z(self._Q), z(self._Q)
        self.mw, self.vw, self.mwu, self.vwu = z(self.w), z(self.w), z(self.wu), z(self.wu)
        self.gP, self.gQ = z(self._P), z(self._Q)
        self.tP = torch.zeros(P.shape[0], dtype=_i32, device=dev)
        self.tQ = torch.zeros(Q.shape[0], dtype=_i32, device=dev)
        self.adam_pow = torch.tensor([hyper.beta1, hyper.beta2], dtype=_f32, device=dev)
        self.losses = torch.zeros(3, dtype=_f32, device=dev)
        self.batch_cap = 0
        self.ws = None
        self.pending_B = 0
        self.pending_kind = LOSS_RUBIBCEBOTH
        self.reserve(batch_cap)

    def reserve(self, B):
        if B > self.batch_cap:
            self.flush()                 # a pending pass keeps lr_t and partial rows in the old workspace
            nbytes = _lib.lib().macr_mf_train_workspace_bytes(B, self.d)
            if nbytes == 0:
                raise MacrError(_lib.E_UNSUPPORTED, "embed_size %d not in {32,64,128,256}" % self.d)
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self._P.device)
            self.batch_cap = B

    _TABLE_NAMES = ("_P", "_Q", "w", "wu", "mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu", "gP", "gQ", "tP", "tQ")

    # P and Q as READERS see them: inside a deferred sequence with the lazy Adam pass individual rows are between 0 and K steps
    # behind (not merely one step old, inconsistent row to row), so reading a table attribute brings every row to the last step
    # first.  The kernels and this class use the raw `_P` / `_Q`.  (A deferred sequence with the DENSE pass leaves the tables one
    # whole step old until flush(), as documented at step().)
    @property
    def P(self):
        if self._seq_lazy is not None and self.pending_B:
            self.flush()
        return self._P

    @P.setter
    def P(self, value):
        self._P = value

    @property
    def Q(self):
        if self._seq_lazy is not None and self.pending_B:
            self.flush()
        return self._Q

    @Q.setter
    def Q(self, value):
        self._Q = value

    def __setattr__(self, name, value):
        # assigning one of the tables (state.P = ...) drops the cached raw pointers
        if name in self._TABLE_NAMES or name == "adam_pow":
            object.__setattr__(self, "_tab_ptrs", None)
        object.__setattr__(self, name, value)

    def invalidate(self):
        """Forget the cached raw pointers.  Needed only after a table's STORAGE was swapped behind the same tensor object
        (`t.data = ...`, `t.set_(...)`, `t.resize_(...)`); assigning a new tensor to an attribute is noticed by itself."""
        self._tab_ptrs = None

    def _tables(self):
        """the 16 table pointers of the C calls, rebuilt only when one of the tensors was replaced (a step is ~36 us:
        sixteen data_ptr() round trips per call were 8 us of it)"""
        if getattr(self, "_tab_ptrs", None) is None:
            object.__setattr__(self, "_tab_ptrs", tuple(_ptr(getattr(self, n)) for n in self._TABLE_NAMES))
            self._dev_index = self._P.device.index if self._P.device.index is not None else torch.cuda.current_device()
            self._pow_ptr = _ptr(self.adam_pow)
        return self._tab_ptrs

    def step(self, kind, u, i, j, losses=None, defer=False):
        """One training step; u,i,j int32 device tensors.  Returns the (3,) device loss tensor.
        defer=True (rubibceboth, rubibce): leave the dense Adam pass pending so that the next step runs it under its
        (B,B) kernel (include/macr_hip.h, MACR_STEP_DEFER); call flush() before reading the parameters."""
        B = u.numel()
        if self.pending_B and self.pending_B != B:
            self.flush()
        self.reserve(B)
        out = self.losses if losses is None else losses
        defer = bool(defer) and kind != LOSS_NORMALBCE          # the (B,B) losses hide the Adam pass under their (B,B) kernel
        if self.pending_B and kind != self.pending_kind:
            self.flush()
        flags = (_lib.STEP_DEFER if defer else 0) | (_lib.STEP_PENDING if self.pending_B else 0)
        tabs = self._tables()
        if not self.pending_B:                  # a deferred sequence starts (or a single complete step): its form
            self._seq_lazy = self._lazy_struct(B) if defer else None
        if self._seq_lazy is not None:
            check(_lib.lib().macr_mf_train_step_lazy(
                kind, B, self.d, self._P.shape[0], self._Q.shape[0], _ptr(u, _i32), _ptr(i, _i32), _ptr(j, _i32),
                *tabs, self._pow_ptr, ctypes.byref(self.hyper),
                _ptr(out, _f32), flags, ctypes.byref(self._seq_lazy), _ptr(self.ws), self.ws.numel(), _stream(self._dev_index)))
        else:
            check(_lib.lib().macr_mf_train_step(
                kind, B, self.d, self._P.shape[0], self._Q.shape[0], _ptr(u, _i32), _ptr(i, _i32), _ptr(j, _i32),
                *tabs, self._pow_ptr, ctypes.byref(self.hyper),
                _ptr(out, _f32), flags, _ptr(self.ws), self.ws.numel(), _stream(self._dev_index)))
        self.pending_B = B if defer else 0
        self.pending_kind = kind
        return out

    def _lazy_struct(self, B):
        """struct macr_lazy_adam for a deferred sequence of batches of B triples, or None (the dense pass every step)"""
        k = self.lazy_period if self.lazy_period is not None else mf_lazy_period_for(self._P.shape[0] + self._Q.shape[0], self.d, B)
        if k <= 1:
            return None
        if self._lazy_bufs is None:
            dev = self._P.device
            self._lazy_bufs = (torch.zeros(_lib.LAZY_STATE_BYTES, dtype=torch.uint8, device=dev),
                               torch.zeros(self._P.shape[0], dtype=_i32, device=dev), torch.zeros(self._Q.shape[0], dtype=_i32, device=dev))
        st, sp, sq = self._lazy_bufs
        return _lib.LazyAdam(_ptr(st), _ptr(sp), _ptr(sq), int(k))

    def flush(self):
        """Complete a pending dense Adam pass (no-op when nothing is pending); a lazy sequence also brings every row to its
        last step: P, Q and the slots are then what the dense pass leaves."""
        if self.pending_B:
            if self._seq_lazy is not None:
                check(_lib.lib().macr_mf_train_flush_lazy(
                    self.pending_kind, self.pending_B, self.d, self._P.shape[0], self._Q.shape[0], *self._tables(),
                    ctypes.byref(self.hyper), ctypes.byref(self._seq_lazy), _ptr(self.ws), self.ws.numel(), _stream(self._dev_index)))
            else:
                check(_lib.lib().macr_mf_train_flush(
                    self.pending_kind, self.pending_B, self.d, self._P.shape[0], self._Q.shape[0], *self._tables(),
                    ctypes.byref(self.hyper), _ptr(self.ws), self.ws.numel(), _stream(self._dev_index)))
            self.pending_B = 0


class LGCNState(object):
    """Device state of LightGCN: ego table T=[user_embedding; item_embedding] (LightGCN.py:226-232),
    branch vectors, Adam slots, the `pre` adjacency (utility/load_data.py:112-121)."""

    def __init__(self, T, n_users, n_items, w, wu, adj, n_layers, hyper, batch_cap, adj_t=None):
        """adj_t: the TRANSPOSED adjacency (CSR) of an asymmetric --adj_type (norm / gcmc / mean: D^-1 A); the backward
        propagation runs on it.  None: the adjacency is symmetric (pre, plain)."""
        _require_f32(T=T, w=w, wu=wu)
        self.T = T.contiguous()
        self.n_users, self.n_items, self.n_layers = n_users, n_items, n_layers
        self.w, self.wu = w.reshape(-1).contiguous(), wu.reshape(-1).contiguous()
        self.adj, self.hyper, self.d = adj, hyper, T.shape[1]
        self.adj_t = adj_t
        z = torch.zeros_like
        self.mT, self.vT = z(self.T), z(self.T)
        self.mw, self.vw, self.mwu, self.vwu = z(self.w), z(self.w), z(self.wu), z(self.wu)
        self.adam_pow = torch.tensor([hyper.beta1, hyper.beta2], dtype=_f32, device=T.device)
        self.losses = torch.zeros(3, dtype=_f32, device=T.device)
        self.batch_cap, self.ws = 0, None
        self.reserve(batch_cap)
        self._E = None

    def reserve(self, B):
        if B > self.batch_cap:
            if self.adj_t is None:
                nbytes = _lib.lib().macr_lgcn_train_workspace_bytes(B, self.T.shape[0], self.d, self.adj._plan_ptrs()[1])
            else:
                nbytes = _lib.lib().macr_lgcn_train_workspace_bytes_t(B, self.T.shape[0], self.d, self.adj._plan_ptrs()[1],
                                                                      self.adj_t._plan_ptrs()[1])
            if nbytes == 0:
                raise MacrError(_lib.E_UNSUPPORTED, "embed_size %d not in {32,64,128,256}" % self.d)
            # zero once (include/macr_hip.h): row flags and the hub rows' arrival counters are
            # zero between steps -- every step leaves them that way
            self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.T.device)
            self.batch_cap = B

    def step(self, kind, u, i, j, losses=None, loss_only=False, dense_layers=False):
        """One training step; loss_only=True computes {loss, mf_loss, emb_loss} of the batch and updates nothing
        (the reference's "test loss" pass, LightGCN.py:799-819).  dense_layers=True: every propagation layer dense
        (default: batch-row-sparse last forward / first backward layer, same result up to summation order)."""
        B = u.numel()
        self.reserve(B)
        out = self.losses if losses is None else losses
        if not loss_only:
            self._E = None
        pd, ph = self.adj._plan_ptrs()
        at = self.adj if self.adj_t is None else self.adj_t
        ptd, pth = at._plan_ptrs()
        check(_lib.lib().macr_lgcn_train_step_t(
            kind, B, self.d, self.n_users, self.n_items, self.n_layers, _ptr(self.adj.ptr, _i32),
            _ptr(self.adj.idx, _i32), _ptr(self.adj.val, _f32), pd, ph,
            _ptr(at.ptr, _i32), _ptr(at.idx, _i32), _ptr(at.val, _f32), ptd, pth, _ptr(u, _i32), _ptr(i, _i32), _ptr(j, _i32),
            _ptr(self.T), _ptr(self.w), _ptr(self.wu), _ptr(self.mT), _ptr(self.vT), _ptr(self.mw), _ptr(self.vw),
            _ptr(self.mwu), _ptr(self.vwu), _ptr(self.adam_pow), ctypes.byref(self.hyper), _ptr(out, _f32),
            (_lib.STEP_LOSS_ONLY if loss_only else 0) | (_lib.STEP_DENSE_LAYERS if dense_layers else 0), _ptr(self.ws),
            self.ws.numel(), _stream()))
        return out

    def propagated(self):
        """ua_embeddings / ia_embeddings (LightGCN.py:130) -- cached until the next step."""
        if self._E is None:
            # persistent buffers: the evaluator replays its launches as a graph keyed on tensor addresses
            if getattr(self, "_E_buf", None) is None:
                self._E_buf = torch.empty_like(self.T)
                self._E_work = torch.zeros(_lib.lib().macr_lgcn_work_floats(self.T.shape[0], self.d, self.adj._plan_ptrs()[1]),
                                           dtype=_f32, device=self.T.device)
            self._E = lgcn_propagate(self.adj, self.T, self.n_layers, out=self._E_buf, work=self._E_work)
        return self._E
