#!/usr/bin/env python3
"""G10: golden losses / gradients / score matrices of the MODEL half, computed by the reference's own graph code.

TensorFlow 1.14 is not installable here, so the reference's graph-BUILDING code
    macr_mf/model.py       BPRMF.create_bce_loss (:277-287), create_bce_loss_two_brach (:158-183),
                           create_bce_loss_two_brach_both (:185-222), create_bpr_loss_two_brach (:124-156, for the
                           rubi_ratings / direct_minus_ratings tensors it defines at :141-142)
    macr_lightgcn/LightGCN.py  LightGCN._create_lightgcn_embed (:288-309) with _split_A_hat (:257-269) and
                           _convert_sp_mat_to_sp_tensor (:537-540), create_bce_loss (:415-429),
                           create_bce_loss_two_brach_both (:495-532)
is imported from /root/reference and executed, unbound, on injected tensors, with a functional stand-in for the
`tensorflow` module: every tf op the code calls (about 20: reduce_sum/mean, multiply, matmul, nn.sigmoid, log,
negative, nn.l2_loss, transpose, squeeze, constant, nn.embedding_lookup, concat, stack, split, SparseTensor,
sparse_tensor_dense_matmul) is mapped to the torch op of the same definition, evaluated eagerly.  Gradients come
from torch.autograd of THAT execution (embedding_lookup -> index_select, whose backward sums duplicate indices: the
IndexedSlices de-duplication).  What this pins: the loss expressions as the reference wrote them -- including the
(B,)*(B,1) -> (B,B) broadcast of model.py:204-205 -- their gradients, and the test-time score formulas.  What it
does NOT pin (still "[TF-1.14 knowledge]"): tf.train.AdamOptimizer's update rule and epsilon
placement, the Xavier initialiser's stream, Eigen's fp32 sigmoid/log kernels and reduction order.

Every case is run twice, in float32 (the reference's dtype) and in float64 (the value the fp32 run approximates).
Only inputs and outputs are stored (tests/golden/G10_model_steps.npz); no reference text travels.

Usage:  python tests/golden/make_golden_model.py
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# ------------------------------------------------------------------------------------------------ tf stand-in
def make_tf():
    tf = types.ModuleType("tensorflow")
    tf.float32, tf.int32 = torch.float32, torch.int32

    def _axis(kw, args):
        if "axis" in kw:
            return kw["axis"]
        if "reduction_indices" in kw:
            return kw["reduction_indices"]
        return args[0] if args else None

    def reduce_sum(x, *args, **kw):
        ax = _axis(kw, args)
        return x.sum() if ax is None else x.sum(dim=ax, keepdim=kw.get("keepdims", False))

    def reduce_mean(x, *args, **kw):
        ax = _axis(kw, args)
        return x.mean() if ax is None else x.mean(dim=ax, keepdim=kw.get("keepdims", False))

    def matmul(a, b, transpose_a=False, transpose_b=False):
        a = a.t() if transpose_a else a
        b = b.t() if transpose_b else b
        return a @ b

    tf.reduce_sum, tf.reduce_mean, tf.matmul = reduce_sum, reduce_mean, matmul
    tf.multiply = lambda a, b: a * b
    tf.log = torch.log
    tf.negative = lambda x: -x
    tf.transpose = lambda x: x.t()
    tf.squeeze = lambda x: x.squeeze()
    tf.concat = lambda xs, axis: torch.cat(list(xs), dim=axis)
    tf.stack = lambda xs, axis=0: torch.stack(list(xs), dim=axis)
    tf.split = lambda x, sizes, axis=0: torch.split(x, list(sizes), dim=axis)
    tf.constant = lambda v, dtype=None, shape=None: torch.full(tuple(shape or ()), float(v))
    tf.zeros = lambda shape: torch.zeros(tuple(shape))
    tf.ones = lambda shape: torch.ones(tuple(shape))

    class SparseTensor(object):
        def __init__(self, indices, values, dense_shape):
            idx = np.asarray(indices).T.astype(np.int64)
            self.indices, self.values, self.shape = idx, np.asarray(values), tuple(int(s) for s in dense_shape)

        def to(self, dtype):
            return torch.sparse_coo_tensor(torch.from_numpy(self.indices), torch.from_numpy(self.values).to(dtype),
                                           self.shape).coalesce()

    tf.SparseTensor = SparseTensor
    tf.sparse_tensor_dense_matmul = lambda a, b: torch.sparse.mm(a.to(b.dtype), b)
    nn = types.ModuleType("tensorflow.nn")
    nn.sigmoid = torch.sigmoid
    nn.l2_loss = lambda x: (x * x).sum() / 2            # tf.nn.l2_loss: sum(t ** 2) / 2
    nn.embedding_lookup = lambda table, ids: table.index_select(0, ids)
    tf.nn = nn
    # session plumbing LightGCN.py touches at import time (:23-31); never used for arithmetic
    tf.ConfigProto = lambda: types.SimpleNamespace(gpu_options=types.SimpleNamespace(allow_growth=False))
    tf.Session = lambda config=None: None
    return tf


def install():
    tf = make_tf()
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.nn"] = tf.nn
    sys.modules["tensorflow.python"] = types.ModuleType("tensorflow.python")
    client = types.ModuleType("tensorflow.python.client")
    client.device_lib = types.SimpleNamespace(list_local_devices=lambda: [])
    sys.modules["tensorflow.python.client"] = client
    if not hasattr(np, "mat"):                  # removed in NumPy 2.0; LightGCN.py:539 uses it
        np.mat = np.asmatrix
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)
    return tf


# ------------------------------------------------------------------------------------------------ inputs
def mf_problem(seed, n_users, n_items, d, B, scale):
    rs = np.random.RandomState(seed)
    P = (rs.standard_normal((n_users, d)) * scale).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * scale).astype(np.float32)
    w = (rs.standard_normal((d, 1)) * 0.3).astype(np.float32)
    wu = (rs.standard_normal((d, 1)) * 0.3).astype(np.float32)
    u = rs.choice(n_users, B, replace=B > n_users).astype(np.int64)
    i = rs.randint(0, n_items, B).astype(np.int64)
    j = rs.randint(0, n_items, B).astype(np.int64)
    i[: B // 3] = 0                                    # hot item: duplicate rows in the gather
    return P, Q, w, wu, u, i, j


def run_mf_loss(BPRMF, fn_name, prob, hyper, dtype):
    """Executes the reference's loss builder; returns losses, gradients and the score tensors it defines."""
    P, Q, w, wu, u, i, j = prob
    t = lambda a: torch.tensor(a, dtype=dtype, requires_grad=True)
    Pt, Qt, wt, wut = t(P), t(Q), t(w), t(wu)
    tf = sys.modules["tensorflow"]
    users = tf.nn.embedding_lookup(Pt, torch.from_numpy(u))               # model.py:35-37
    pos = tf.nn.embedding_lookup(Qt, torch.from_numpy(i))
    neg = tf.nn.embedding_lookup(Qt, torch.from_numpy(j))
    me = types.SimpleNamespace(w=wt, w_user=wut, alpha=hyper["alpha"], beta=hyper["beta"], decay=hyper["decay"],
                               batch_size=hyper["batch_size"], rubi_c=torch.zeros(1, dtype=dtype),
                               batch_ratings=tf.matmul(users, pos, transpose_a=False, transpose_b=True))   # :45
    mf_loss, reg_loss = getattr(BPRMF, fn_name)(me, users, pos, neg)
    loss = mf_loss + reg_loss                                             # :73 / :94
    grads = torch.autograd.grad(loss, [Pt, Qt, wt, wut], allow_unused=True)
    z = lambda g, ref: np.zeros_like(ref) if g is None else g.detach().numpy()
    out = {"loss": float(loss.detach()), "mf_loss": float(mf_loss.detach()), "reg_loss": float(reg_loss.detach()),
           "dP": z(grads[0], P), "dQ": z(grads[1], Q), "dw": z(grads[2], w), "dwu": z(grads[3], wu)}
    for name in ("mf_loss_ori", "mf_loss_item", "mf_loss_user"):
        if hasattr(me, name):
            out[name] = float(getattr(me, name).detach())
    return out


def run_mf_scores(BPRMF, prob, c, dtype):
    """Test-time score tensors (model.py:45, :141-142, :199-201) for ALL users x ALL items: the builders are called
    with users = P, pos_items = Q, which needs n_users == n_items (the loss they also build is discarded)."""
    P, Q, w, wu = prob[:4]
    assert P.shape[0] == Q.shape[0]
    t = lambda a: torch.tensor(a, dtype=dtype)
    Pt, Qt, wt, wut = t(P), t(Q), t(w), t(wu)
    tf = sys.modules["tensorflow"]
    me = types.SimpleNamespace(w=wt, w_user=wut, alpha=0.0, beta=0.0, decay=0.0, batch_size=1,
                               rubi_c=c * torch.ones(1, dtype=dtype),     # update_c, model.py:313
                               batch_ratings=tf.matmul(Pt, Qt, transpose_a=False, transpose_b=True))
    BPRMF.create_bce_loss_two_brach_both(me, Pt, Qt, Qt)
    both = {k: getattr(me, k).detach().numpy() for k in ("rubi_ratings_both", "direct_minus_ratings_both")}
    BPRMF.create_bpr_loss_two_brach(me, Pt, Qt, Qt)
    both.update({k: getattr(me, k).detach().numpy() for k in ("rubi_ratings", "direct_minus_ratings")})
    both["batch_ratings"] = me.batch_ratings.detach().numpy()
    return both


def lgcn_problem(seed, n_users, n_items, d, B, n_inter):
    rs = np.random.RandomState(seed)
    R = sp.dok_matrix((n_users, n_items), dtype=np.float32)
    for _ in range(n_inter):
        R[int(rs.randint(n_users)), int(rs.zipf(1.3) % n_items)] = 1.0
    for uu in range(n_users):                     # every user and item has a neighbour (finite D^-1/2)
        R[uu, int(rs.randint(n_items))] = 1.0
    for ii in range(n_items):
        R[int(rs.randint(n_users)), ii] = 1.0
    R = R.tocsr()
    N = n_users + n_items
    A = sp.lil_matrix((N, N), dtype=np.float32)
    A[:n_users, n_users:] = R
    A[n_users:, :n_users] = R.T
    A = A.tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    dinv = np.power(deg, -0.5)
    pre = sp.diags(dinv).dot(A).dot(sp.diags(dinv)).tocsr().astype(np.float32)    # utility/load_data.py:112-121
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w = (rs.standard_normal((d, 1)) * 0.3).astype(np.float32)
    wu = (rs.standard_normal((d, 1)) * 0.3).astype(np.float32)
    u = rs.choice(n_users, B, replace=B > n_users).astype(np.int64)
    i = (rs.zipf(1.3, B) % n_items).astype(np.int64)
    j = rs.randint(0, n_items, B).astype(np.int64)
    return pre, P, Q, w, wu, u, i, j


def run_lgcn(LightGCN, fn_name, prob, hyper, n_layers, dtype):
    pre, P, Q, w, wu, u, i, j = prob
    t = lambda a: torch.tensor(a, dtype=dtype, requires_grad=True)
    Pt, Qt, wt, wut = t(P), t(Q), t(w), t(wu)
    tf = sys.modules["tensorflow"]
    me = types.SimpleNamespace(n_users=P.shape[0], n_items=Q.shape[0], n_fold=100, norm_adj=pre, n_layers=n_layers,
                               node_dropout_flag=0, weights={"user_embedding": Pt, "item_embedding": Qt},
                               w=wt, w_user=wut, alpha=hyper["alpha"], beta=hyper["beta"], decay=hyper["decay"],
                               batch_size=hyper["batch_size"], rubi_c=torch.zeros(1, dtype=dtype))
    me._convert_sp_mat_to_sp_tensor = lambda X: LightGCN._convert_sp_mat_to_sp_tensor(me, X)
    me._split_A_hat = lambda X: LightGCN._split_A_hat(me, X)
    ua, ia = LightGCN._create_lightgcn_embed(me)                          # LightGCN.py:130
    ui, ii, ji = torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j)
    ug, pg, ng = (tf.nn.embedding_lookup(ua, ui), tf.nn.embedding_lookup(ia, ii), tf.nn.embedding_lookup(ia, ji))   # :145-147
    me.u_g_embeddings_pre = tf.nn.embedding_lookup(Pt, ui)                # :148-150
    me.pos_i_g_embeddings_pre = tf.nn.embedding_lookup(Qt, ii)
    me.neg_i_g_embeddings_pre = tf.nn.embedding_lookup(Qt, ji)
    me.batch_ratings = tf.matmul(ug, pg, transpose_a=False, transpose_b=True)     # :166
    mf_loss, emb_loss, _ = getattr(LightGCN, fn_name)(me, ug, pg, ng)
    loss = mf_loss + emb_loss                                             # :185 / :200
    grads = torch.autograd.grad(loss, [Pt, Qt, wt, wut], allow_unused=True)
    z = lambda g, ref: np.zeros_like(ref) if g is None else g.detach().numpy()
    return {"loss": float(loss.detach()), "mf_loss": float(mf_loss.detach()), "emb_loss": float(emb_loss.detach()),
            "ua": ua.detach().numpy(), "ia": ia.detach().numpy(),
            "dP": z(grads[0], P), "dQ": z(grads[1], Q), "dw": z(grads[2], w), "dwu": z(grads[3], wu)}


# ------------------------------------------------------------------------------------------------ main
def main():
    install()
    sys.path.insert(0, os.path.join(REF, "macr_mf"))
    sys.argv = ["make_golden_model"]
    import importlib
    mf_model = importlib.import_module("model")
    BPRMF = mf_model.BPRMF

    out = {}
    hyper = dict(alpha=1e-2, beta=1e-3, decay=1e-5, batch_size=1024)
    cases = {"a": (11, 60, 40, 32, 48, 0.3), "b": (12, 300, 80, 64, 257, 0.6), "c": (13, 90, 50, 32, 96, 1.5)}
    for tag, (seed, nu, ni, d, B, scale) in cases.items():
        prob = mf_problem(seed, nu, ni, d, B, scale)
        for k, v in zip(("P", "Q", "w", "wu", "u", "i", "j"), prob):
            out["mf_%s/%s" % (tag, k)] = v
        for loss_name, fn in (("normalbce", "create_bce_loss"), ("rubibce", "create_bce_loss_two_brach"),
                              ("rubibceboth", "create_bce_loss_two_brach_both")):
            for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
                res = run_mf_loss(BPRMF, fn, prob, hyper, dt)
                for k, v in res.items():
                    out["mf_%s/%s/%s/%s" % (tag, loss_name, dname, k)] = np.asarray(v)
    # test-time scores, all users x all items (square case), c = 0 and the README's c = 40
    prob = mf_problem(21, 64, 64, 32, 8, 0.4)
    for k, v in zip(("P", "Q", "w", "wu"), prob[:4]):
        out["mf_scores/%s" % k] = v
    for c in (0.0, 40.0):
        for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
            for k, v in run_mf_scores(BPRMF, prob, c, dt).items():
                out["mf_scores/c%g/%s/%s" % (c, dname, k)] = v

    # LightGCN: the class is taken from the module source without running its import-time harness
    # (utility.batch_test parses the command line and loads a dataset): compile the module with that import stubbed.
    sys.path.insert(0, os.path.join(REF, "macr_lightgcn"))
    helper = types.ModuleType("utility.helper")
    bt = types.ModuleType("utility.batch_test")
    bt.args = types.SimpleNamespace(gpu_id=0)
    helper.np = bt.np = np                        # the reference gets `np` through these star imports
    util = types.ModuleType("utility")
    sys.modules.update({"utility": util, "utility.helper": helper, "utility.batch_test": bt})
    lg = importlib.import_module("LightGCN")
    LightGCN = lg.LightGCN
    for tag, (seed, nu, ni, d, B, n_inter) in {"a": (31, 70, 45, 32, 64, 400), "b": (32, 260, 150, 64, 200, 1500)}.items():
        prob = lgcn_problem(seed, nu, ni, d, B, n_inter)
        pre = prob[0]
        out["lgcn_%s/indptr" % tag], out["lgcn_%s/indices" % tag], out["lgcn_%s/data" % tag] = pre.indptr, pre.indices, pre.data
        for k, v in zip(("P", "Q", "w", "wu", "u", "i", "j"), prob[1:]):
            out["lgcn_%s/%s" % (tag, k)] = v
        for loss_name, fn in (("bce", "create_bce_loss"), ("bceboth", "create_bce_loss_two_brach_both")):
            for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
                res = run_lgcn(LightGCN, fn, prob, hyper, 2, dt)
                for k, v in res.items():
                    if dname == "f64" or np.ndim(v) == 0:
                        out["lgcn_%s/%s/%s/%s" % (tag, loss_name, dname, k)] = np.asarray(v)
    out["hyper"] = np.asarray([hyper["alpha"], hyper["beta"], hyper["decay"], hyper["batch_size"]], np.float64)
    path = os.path.join(HERE, "G10_model_steps.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays, %.1f KB)" % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
