"""ctypes/NumPy face of oracle/macr_oracle.c -- TEST INFRASTRUCTURE ONLY.

Every function here is a thin marshalling layer; the arithmetic (and the
reference file:line each piece follows) lives in macr_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "macr_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libmacr_oracle.so")
_LIB_FAST = os.path.join(_HERE, "_build", "libmacr_oracle_fast.so")
_REF_LIB = os.path.join(_HERE, "_ref", "libref_eval.so")

LOSS_NORMALBCE, LOSS_RUBIBCEBOTH, LOSS_RUBIBCE = 0, 1, 2
SCORE_NORMAL, SCORE_RUBI_BOTH, SCORE_RUBI, SCORE_DIRECT_MINUS, SCORE_DIRECT_MINUS_BOTH = 0, 1, 2, 3, 4

_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_d = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_ci, _cf, _cz = ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_vp = ctypes.c_void_p


def build(force=False):
    """Compile the C restatement (gcc) and, where /root/reference exists, the
    reference's own C++ evaluator into oracle/_ref."""
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if force or (not os.path.exists(_LIB_FAST)) or os.path.getmtime(_LIB_FAST) < os.path.getmtime(_SRC):
        subprocess.check_call(["make", "-s", "-C", _HERE, "fast"])
    if os.path.isdir("/root/reference/macr_lightgcn/evaluator/cpp/include"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_libs = {}
_use_fast = False


class fast(object):
    """`with oracle.fast(): ...` -- the calls inside go to the speed build of the same C file (Makefile target `fast`:
    -O3, AVX2, -ffast-math).  For bench.py's cpu_baseline leg only; the checker is the strict build."""

    def __enter__(self):
        global _use_fast
        self._prev, _use_fast = _use_fast, True
        return self

    def __exit__(self, *exc):
        global _use_fast
        _use_fast = self._prev


def lib():
    global _lib
    path = _LIB_FAST if _use_fast else _LIB
    if path in _libs:
        return _libs[path]
    if True:
        build()
        L = ctypes.CDLL(path)
        L.orc_gather_rows.argtypes = [_f, _i, _ci, _ci, _f]
        L.orc_scatter_add_rows.argtypes = [_f, _i, _ci, _ci, _f]
        L.orc_pair_loss_grad.argtypes = [_ci, _ci, _ci, _f, _f, _f, _f, _f, _cf, _cf,
                                         _f, _f, _f, _f, _f, _f, _f]
        L.orc_l2_reg.argtypes = [_ci, _ci, _f, _f, _f, _cf, _ci, _vp, _vp, _vp]
        L.orc_l2_reg.restype = _cf
        L.orc_adam_lr_t.argtypes = [_cf, _f]
        L.orc_adam_lr_t.restype = _cf
        L.orc_adam_dense.argtypes = [_f, _f, _f, _vp, _cz, _cf, _cf, _cf, _cf]
        L.orc_mf_train_step.argtypes = [_ci] * 5 + [_i, _i, _i] + [_f] * 13 + [_cf] * 7 + [_ci, _f]
        L.orc_spmm_csr.argtypes = [_ci, _ci, _i, _i, _f, _f, _f]
        L.orc_lgcn_propagate.argtypes = [_ci, _ci, _ci, _i, _i, _f, _f, _f, _f]
        L.orc_lgcn_train_step.argtypes = [_ci] * 6 + [_i, _i, _f, _i, _i, _i] + [_f] * 10 + \
            [_cf] * 7 + [_ci, _f]
        L.orc_lgcn_train_step_t.argtypes = [_ci] * 6 + [_i, _i, _f, _i, _i, _f, _i, _i, _i] + [_f] * 10 + \
            [_cf] * 7 + [_ci, _f]
        L.orc_lgcn_train_step_t.restype = None
        L.orc_branch_sigmoid.argtypes = [_f, _ci, _ci, _f, _f]
        L.orc_score_topk.argtypes = [_ci, _ci, _ci, _ci, _f, _f, _vp, _vp, _cf, _vp, _vp, _ci,
                                     _ci, _ci, _f, _i, _i]
        L.orc_score_matrix.argtypes = [_ci, _ci, _ci, _ci, _f, _f, _vp, _vp, _cf, _f]
        L.orc_score_matrix.restype = None
        L.orc_topk_scores.argtypes = [_ci, _ci, _f, _vp, _vp, _ci, _f, _i, _i]
        L.orc_topk_merge.argtypes = [_ci, _ci, _ci, _f, _i, _f, _i, _i]
        L.orc_metrics_foldout.argtypes = [_ci, _ci, _i, _i, _i, _f]
        L.orc_metrics_mf.argtypes = [_ci, _ci, _i, _vp, _i, _i, _i, _ci, _d]
        L.orc_sample_triples.argtypes = [ctypes.c_uint64, ctypes.c_uint64, _ci, _ci, _vp, _ci, _i, _i, _vp, _vp, _i]
        L.orc_sample_triples.restype = None
        for fn in ("orc_gather_rows", "orc_scatter_add_rows", "orc_pair_loss_grad", "orc_adam_dense",
                   "orc_mf_train_step", "orc_spmm_csr", "orc_lgcn_propagate", "orc_lgcn_train_step",
                   "orc_branch_sigmoid", "orc_score_topk", "orc_topk_scores", "orc_topk_merge", "orc_metrics_foldout",
                   "orc_metrics_mf"):
            getattr(L, fn).restype = None
        _libs[path] = L
        if path == _LIB:
            _lib = L
    return _libs[path]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def csr_from_lists(lists):
    """list-of-lists -> (ptr int32[U+1], idx int32[nnz]) with every row sorted ascending."""
    ptr = np.zeros(len(lists) + 1, np.int32)
    for k, row in enumerate(lists):
        ptr[k + 1] = ptr[k] + len(row)
    idx = np.fromiter((x for row in lists for x in sorted(row)), dtype=np.int32, count=int(ptr[-1]))
    return ptr, idx


# ------------------------------------------------------------------ model step
class AdamState(object):
    """Adam slots for one parameter set + TF's fp32 beta powers (t starts at 1)."""

    def __init__(self, shapes, b1=0.9, b2=0.999):
        self.m = [np.zeros(s, np.float32) for s in shapes]
        self.v = [np.zeros(s, np.float32) for s in shapes]
        self.power = np.asarray([b1, b2], np.float32)


def pair_loss_grad(kind, eu, ei, ej, w, wu, alpha, beta):
    eu, ei, ej, w, wu = map(_f32, (eu, ei, ej, np.ravel(w), np.ravel(wu)))
    B, d = eu.shape
    parts = np.zeros(4, np.float32)
    deu, dei, dej = (np.zeros((B, d), np.float32) for _ in range(3))
    dw, dwu = np.zeros(d, np.float32), np.zeros(d, np.float32)
    fwd = np.zeros((5, B), np.float32)
    lib().orc_pair_loss_grad(kind, B, d, eu, ei, ej, w, wu, alpha, beta, parts, deu, dei, dej, dw, dwu, fwd)
    return dict(mf=parts[0], l_ori=parts[1], l_item=parts[2], l_user=parts[3],
                deu=deu, dei=dei, dej=dej, dw=dw, dwu=dwu, fwd=fwd)


def l2_reg(eu, ei, ej, decay, batch_size_cfg):
    eu, ei, ej = map(_f32, (eu, ei, ej))
    B, d = eu.shape
    return float(lib().orc_l2_reg(B, d, eu, ei, ej, decay, batch_size_cfg, None, None, None))


def mf_train_step(kind, u, i, j, P, Q, w, wu, st, lr, decay, alpha, beta, batch_size_cfg,
                  b1=0.9, b2=0.999, eps=1e-8):
    """In-place step on float32 arrays P,Q,w,wu; st = AdamState([P.shape,Q.shape,(d,),(d,)])."""
    u, i, j = map(_i32, (u, i, j))
    B, d = len(u), P.shape[1]
    losses = np.zeros(3, np.float32)
    lib().orc_mf_train_step(kind, B, d, P.shape[0], Q.shape[0], u, i, j, P, Q, w, wu,
                            st.m[0], st.v[0], st.m[1], st.v[1], st.m[2], st.v[2], st.m[3], st.v[3],
                            st.power, lr, b1, b2, eps, decay, alpha, beta, batch_size_cfg, losses)
    return losses


def spmm_csr(indptr, indices, data, X):
    indptr, indices, data, X = _i32(indptr), _i32(indices), _f32(data), _f32(X)
    Y = np.empty_like(X)
    lib().orc_spmm_csr(len(indptr) - 1, X.shape[1], indptr, indices, data, X, Y)
    return Y


def lgcn_propagate(indptr, indices, data, E0, n_layers):
    indptr, indices, data, E0 = _i32(indptr), _i32(indices), _f32(data), _f32(E0)
    N, d = E0.shape
    E = np.empty_like(E0)
    work = np.empty((2, N, d), np.float32)
    lib().orc_lgcn_propagate(N, d, n_layers, indptr, indices, data, E0, E, work)
    return E


def lgcn_train_step(kind, n_users, n_items, n_layers, indptr, indices, data, u, i, j, T, w, wu, st,
                    lr, decay, alpha, beta, batch_size_cfg, b1=0.9, b2=0.999, eps=1e-8, transposed=None):
    """In-place step on T=[P;Q] (N,d), w, wu; st = AdamState([T.shape,(d,),(d,)]).
    transposed = (indptr, indices, data) of A^T in CSR for an asymmetric adjacency (--adj_type norm / gcmc / mean); None: A^T = A."""
    u, i, j = map(_i32, (u, i, j))
    losses = np.zeros(3, np.float32)
    tp, ti, td = (indptr, indices, data) if transposed is None else transposed
    lib().orc_lgcn_train_step_t(kind, len(u), T.shape[1], n_users, n_items, n_layers,
                              _i32(indptr), _i32(indices), _f32(data), _i32(tp), _i32(ti), _f32(td), u, i, j, T, w, wu,
                              st.m[0], st.v[0], st.m[1], st.v[1], st.m[2], st.v[2], st.power,
                              lr, b1, b2, eps, decay, alpha, beta, batch_size_cfg, losses)
    return losses


# ------------------------------------------------------------------ evaluator
def branch_sigmoid(rows, w):
    rows, w = _f32(rows), _f32(np.ravel(w))
    out = np.empty(rows.shape[0], np.float32)
    lib().orc_branch_sigmoid(rows, rows.shape[0], rows.shape[1], w, out)
    return out


def score_topk(kind, Urows, Irows, K, sig_u=None, sig_i=None, c=0.0, mask=None, item_offset=0,
               fill_masked=False):
    """mask = (ptr, idx) CSR of per-user sorted global item ids to exclude."""
    Urows, Irows = _f32(Urows), _f32(Irows)
    U, d = Urows.shape
    N = Irows.shape[0]
    su = _f32(sig_u) if sig_u is not None else None
    si = _f32(sig_i) if sig_i is not None else None
    mp, mi = (_i32(mask[0]), _i32(mask[1])) if mask is not None else (None, None)
    val = np.empty((U, K), np.float32)
    idx = np.empty((U, K), np.int32)
    cnt = np.empty(U, np.int32)
    lib().orc_score_topk(kind, U, N, d, Urows, Irows, _ptr(su), _ptr(si), c, _ptr(mp), _ptr(mi),
                         item_offset, K, int(fill_masked), val, idx, cnt)
    return val, idx, cnt


def score_matrix(kind, Urows, Irows, sig_u=None, sig_i=None, c=0.0):
    """Dense (U,N) test-time scores of the given kind (model.py:45, :141-142, :199-201)."""
    Urows, Irows = _f32(Urows), _f32(Irows)
    U, d = Urows.shape
    N = Irows.shape[0]
    su = _f32(sig_u) if sig_u is not None else None
    si = _f32(sig_i) if sig_i is not None else None
    out = np.empty((U, N), np.float32)
    lib().orc_score_matrix(kind, U, N, d, Urows, Irows, _ptr(su), _ptr(si), c, out)
    return out


def topk_scores(scores, K, mask=None):
    """Top-K of a (U,N) score matrix, ties -> lower id; optional candidate mask CSR."""
    scores = _f32(scores)
    U, N = scores.shape
    mp, mi = (_i32(mask[0]), _i32(mask[1])) if mask is not None else (None, None)
    val, idx, cnt = np.empty((U, K), np.float32), np.empty((U, K), np.int32), np.empty(U, np.int32)
    lib().orc_topk_scores(U, N, scores, _ptr(mp), _ptr(mi), K, val, idx, cnt)
    return val, idx, cnt


def topk_merge(vals, idxs):
    vals, idxs = _f32(vals), _i32(idxs)
    W, U, K = vals.shape
    ov, oi, oc = np.empty((U, K), np.float32), np.empty((U, K), np.int32), np.empty(U, np.int32)
    lib().orc_topk_merge(W, U, K, vals, idxs, ov, oi, oc)
    return ov, oi, oc


def metrics_foldout(rankings, gt):
    """rankings (U,K) int32; gt = (ptr, idx) sorted CSR -> (U,5K) float32 [prec|recall|ap|ndcg|mrr]."""
    rankings = _i32(rankings)
    U, K = rankings.shape
    out = np.zeros((U, 5 * K), np.float32)
    lib().orc_metrics_foldout(U, K, rankings, _i32(gt[0]), _i32(gt[1]), out)
    return out


def metrics_mf(rankings, cnt, gt, Ks):
    """-> (U,4,len(Ks)) float64: precision, recall, ndcg, hit_ratio per user."""
    rankings = _i32(rankings)
    U, Kmax = rankings.shape
    Ks = _i32(Ks)
    out = np.zeros((U, 4, len(Ks)), np.float64)
    c = _i32(cnt) if cnt is not None else None
    lib().orc_metrics_mf(U, Kmax, rankings, _ptr(c), _i32(gt[0]), _i32(gt[1]), Ks, len(Ks), out)
    return out


# ------------------------------------------------------------------ reference C++ evaluator (oracle/_ref)
def sample_triples(seed, step, B, n_items, train, pool=None, exclude=None, n_pool=None):
    """Batch `step` of the device sampler's stream (macr_amd/csrc/sample_kernels.hip) -> (3, B) int32 users, positives,
    negatives.  train / exclude: (ptr, idx) CSR of ascending item lists per user id; pool: user ids to draw from."""
    tp, ti = _i32(train[0]), _i32(train[1])
    pool = None if pool is None else _i32(pool)
    n_pool = (len(tp) - 1 if pool is None else len(pool)) if n_pool is None else n_pool
    ex = (None, None) if exclude is None else (_i32(exclude[0]), _i32(exclude[1]))
    out = np.empty((3, B), np.int32)
    lib().orc_sample_triples(int(seed), int(step), B, n_items, _ptr(pool), n_pool, tp, ti if len(ti) else np.zeros(1, np.int32),
                             _ptr(ex[0]), _ptr(ex[1]), out)
    return out


def have_ref():
    return os.path.exists(_REF_LIB)


_ref = None


def ref_eval_score_matrix_foldout(score_matrix, test_items, top_k=20, thread_num=4):
    """The reference's own C++ evaluator (tools.h:24 + evaluate_foldout.h:115), compiled from
    /root/reference by oracle/Makefile.  Returns (results (U,5K) float32, rankings (U,K) int32)."""
    global _ref
    if _ref is None:
        R = ctypes.CDLL(_REF_LIB)
        R.ref_top_k_array_index.argtypes = [_f, _ci, _ci, _ci, _ci, _i]
        R.ref_top_k_array_index.restype = None
        R.ref_evaluate_foldout.argtypes = [_ci, _i, _ci, ctypes.POINTER(ctypes.POINTER(_ci)), _i, _ci, _f]
        R.ref_evaluate_foldout.restype = None
        _ref = R
    scores = _f32(score_matrix)
    U, N = scores.shape
    rank = np.zeros((U, top_k), np.int32)
    _ref.ref_top_k_array_index(scores, N, U, top_k, thread_num, rank)
    gts = [_i32(t) for t in test_items]
    ptrs = (ctypes.POINTER(_ci) * U)(*[t.ctypes.data_as(ctypes.POINTER(_ci)) for t in gts])
    lens = _i32([len(t) for t in gts])
    res = np.zeros((U, 5 * top_k), np.float32)
    _ref.ref_evaluate_foldout(U, rank, top_k, ptrs, lens, thread_num, res)
    return res, rank
