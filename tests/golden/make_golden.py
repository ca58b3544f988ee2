#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever read the small .npz/.json files this script leaves behind.
No reference source, bytecode or text is written into the repository: the
fixtures hold inputs (or the seeds that regenerate them) and the outputs the
reference computed for them.

What is imported / executed from the reference:
  G1  macr_mf/load_data.py Data (ctor :504, load_ori_data :26)      loader stats
      macr_lightgcn/utility/load_data.py Data (:15)
  G2  macr_mf/load_data.py Data.sample (:543)                        sampler stream
  G3  macr_lightgcn/utility/load_data.py Data.sample (:174), sample_test (:214)
  G4  macr_lightgcn/utility/load_data.py get_adj_mat (:95)           4 adjacency matrices
  G5  macr_mf/train.py test() (:162) through a stub session          MF evaluator
  G6  macr_lightgcn/utility/batch_test.py test() (:26), stub session LightGCN evaluator
  G7  evaluator/cpp/include/{tools,evaluate_foldout}.h compiled by   raw C++ evaluator
      oracle/Makefile into oracle/_ref/libref_eval.so
  G8  macr_mf/train.py metric functions (:32-117)                    unit cases

TensorFlow is not installed here, so a do-nothing ``tensorflow`` module is put
in sys.modules (the reference's evaluators only *pass* tf objects to
``sess.run``); ``np.asfarray`` (removed in NumPy 2) is shimmed; the Cython
wrapper around the C++ evaluator is replaced by a ctypes call into the same
C++ headers (oracle/_ref).  None of this changes reference arithmetic.

Usage:  python tests/golden/make_golden.py          (runs both halves)
"""
import ctypes
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SEED = 12345


# --------------------------------------------------------------------------
# shared helpers
# --------------------------------------------------------------------------
def sha_ints(obj):
    """sha256 over a canonical text form of {user: [items]} / list-of-lists."""
    h = hashlib.sha256()
    if isinstance(obj, dict):
        for k in sorted(obj):
            h.update(("%d:" % k).encode())
            h.update((",".join(str(int(x)) for x in obj[k]) + ";").encode())
    else:
        for row in obj:
            h.update((",".join(str(int(x)) for x in row) + ";").encode())
    return h.hexdigest()


def sha_arr(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def score_matrix(kind, n_users, n_items, seed):
    """Deterministic score matrices, regenerable from (kind, shape, seed).

    RandomState is NumPy's frozen legacy generator, so the stream is stable
    across NumPy versions."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((n_users, n_items)).astype(np.float32)
    if kind == "normal":
        return x
    if kind == "ties":          # heavy exact ties: 9 distinct values
        return (np.round(x * 2.0) / 2.0).clip(-2, 2).astype(np.float32)
    if kind == "popular":       # popularity-skewed like a trained model
        pop = rs.standard_normal(n_items).astype(np.float32) * 3.0
        return (x + pop[None, :]).astype(np.float32)
    raise ValueError(kind)


def write_tiny_dataset(root):
    """A 40-user x 25-item dataset with the edge cases the reference tolerates:
    users absent from train, a user whose train list leaves < K candidates,
    users absent from test, duplicate-free ragged lists."""
    rs = np.random.RandomState(7)
    n_users, n_items = 40, 25
    train, test = {}, {}
    for u in range(n_users):
        if u in (3, 17):            # test-only users (no train line)
            k = 0
        elif u == 5:                # leaves 25-21 = 4 candidates  (< K=20)
            k = 21
        else:
            k = int(rs.randint(1, 9))
        items = sorted(rs.choice(n_items, size=k, replace=False).tolist()) if k else []
        rest = [i for i in range(n_items) if i not in items]
        if u % 4 != 1:              # a quarter of the users are not tested
            t = int(rs.randint(1, min(4, len(rest)) + 1))
            test[u] = sorted(rs.choice(rest, size=t, replace=False).tolist())
        if items:
            train[u] = items
    # make sure max ids appear so n_users/n_items are pinned
    train[n_users - 1] = sorted(set(train.get(n_users - 1, [])) | {n_items - 1})
    d = os.path.join(root, "tiny")
    os.makedirs(d, exist_ok=True)
    for name, tab in (("train.txt", train), ("test.txt", test)):
        with open(os.path.join(d, name), "w") as f:
            for u in sorted(tab):
                f.write(" ".join([str(u)] + [str(i) for i in tab[u]]) + "\n")
    return d


def install_common_shims():
    tf = types.ModuleType("tensorflow")
    tf.__getattr__ = lambda name: None          # any tf.xyz -> None
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.python"] = types.ModuleType("tensorflow.python")
    client = types.ModuleType("tensorflow.python.client")
    client.device_lib = None
    sys.modules["tensorflow.python.client"] = client
    if not hasattr(np, "asfarray"):             # removed in NumPy 2.0
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)


class StubModel(object):
    """Stands in for BPRMF/LightGCN: every attribute is just its own name."""
    Ks = [20]

    def __getattr__(self, name):
        return name


class StubSession(object):
    """sess.run(fetch, feed) -> rows of a fixed score matrix for feed[users]."""

    def __init__(self, full_scores):
        self.full = full_scores
        self.fetches = []

    def run(self, fetch, feed_dict=None):
        self.fetches.append(fetch)
        users = feed_dict["users"]
        return self.full[np.asarray(list(users), dtype=np.int64)].astype(np.float32).copy()


# --------------------------------------------------------------------------
# MF half  (G1, G2, G5, G8)
# --------------------------------------------------------------------------
def run_mf(dataset, out):
    install_common_shims()
    work = tempfile.mkdtemp(prefix="golden_mf_")
    os.makedirs(os.path.join(work, "data"))
    if dataset == "tiny":
        write_tiny_dataset(os.path.join(work, "data"))
        shutil.copytree(os.path.join(work, "data", "tiny"), os.path.join(HERE, "tiny_data"),
                        dirs_exist_ok=True)
    else:
        os.symlink(os.path.join(REF, "data", dataset), os.path.join(work, "data", dataset))
    os.chdir(work)                               # reference reads ./data/<dataset>/
    sys.path.insert(0, os.path.join(REF, "macr_mf"))
    bs = 1024 if dataset != "tiny" else 16
    sys.argv = ["train.py", "--dataset", dataset, "--batch_size", str(bs),
                "--Ks", "[20]" if dataset != "tiny" else "[5, 20]"]
    import random
    import train as ref_train                    # noqa: E402  (reference module)
    data = ref_train.data
    g = {}

    # ---- G1 loader
    g["G1"] = dict(
        n_users=data.n_users, n_items=data.n_items, n_train=data.n_train,
        n_test=data.n_test, n_test_users=len(data.test_user_list),
        n_train_users=len([u for u in data.train_user_list if data.train_user_list[u]]),
        train_sha=sha_ints({u: v for u, v in data.train_user_list.items() if v}),
        test_sha=sha_ints(dict(data.test_user_list)),
        train_item_sha=sha_ints({i: v for i, v in data.train_item_list.items() if v}),
        test_users_order_sha=sha_arr(np.asarray(list(data.test_user_list.keys()), np.int64)),
    )

    # ---- G2 sampler stream (train.py:333-336 seeds python random)
    random.seed(SEED)
    np.random.seed(SEED)
    batches = [data.sample() for _ in range(3)]
    g2 = np.asarray(batches, dtype=np.int32)     # (3 batches, 3 lists, B)
    np.savez_compressed(os.path.join(HERE, "G2_mf_sampler_%s.npz" % dataset), batches=g2)
    g["G2"] = dict(batch_size=bs, seed=SEED, sha=sha_arr(g2))

    # ---- G5 evaluator through a stub session
    users_to_test = list(data.test_user_list.keys())
    g5 = {}
    kinds = ["normal", "ties", "popular"]
    for kind in kinds:
        full = score_matrix(kind, data.n_users, data.n_items, seed=100 + kinds.index(kind))
        for model_type in ("o", "rubi_both"):
            sess = StubSession(full)
            ret = ref_train.test(sess, StubModel(), users_to_test, model_type=model_type)
            g5["%s/%s" % (kind, model_type)] = {
                k: [float(x) for x in v] for k, v in ret.items()}
            assert set(sess.fetches) == {"batch_ratings" if model_type == "o" else "rubi_ratings_both"}
    g["G5"] = dict(Ks=ref_train.Ks, batch_size=bs, seeds={k: 100 + i for i, k in enumerate(kinds)},
                   results=g5)

    # ---- G8 metric unit cases
    cases = [
        ([1, 0, 0, 1, 0], 5, 2), ([0, 0, 0, 0, 0], 5, 3), ([1, 1, 1], 20, 1),
        ([0, 1] * 10, 20, 30), ([1] * 20, 20, 7), ([0] * 19 + [1], 20, 1), ([1], 20, 4),
    ]
    g8 = []
    for r, k, npos in cases:
        g8.append(dict(
            r=r, k=k, n_pos=npos,
            precision=float(ref_train.precision_at_k(r, k)),
            recall=float(ref_train.recall_at_k(r, k, npos)),
            ndcg=float(ref_train.ndcg_at_k(r, k, npos)),
            hit=float(ref_train.hit_at_k(r, k)),
            dcg=float(ref_train.dcg_at_k(r, k))))
    g["G8"] = g8

    # ---- G9 default values of every CLI flag (macr_mf/parse.py:3-92)
    import parse as ref_parse
    argv, sys.argv = sys.argv, ["train.py"]
    g["G9"] = {k: v for k, v in vars(ref_parse.parse_args()).items()}
    sys.argv = argv
    with open(out, "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    shutil.rmtree(work, ignore_errors=True)


# --------------------------------------------------------------------------
# LightGCN half (G1, G3, G4, G6, G7)
# --------------------------------------------------------------------------
def load_ref_lib():
    lib = ctypes.CDLL(os.path.join(REPO, "oracle", "_ref", "libref_eval.so"))
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    lib.ref_top_k_array_index.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip]
    lib.ref_top_k_array_index.restype = None
    lib.ref_evaluate_foldout.argtypes = [ctypes.c_int, ip, ctypes.c_int,
                                         ctypes.POINTER(ctypes.POINTER(ctypes.c_int)), ip,
                                         ctypes.c_int, fp]
    lib.ref_evaluate_foldout.restype = None
    return lib


def ref_eval_score_matrix_foldout(lib, score_matrix_, test_items, top_k=20, thread_num=None):
    """ctypes re-binding of the reference C ABI (apt_evaluate_foldout.pyx:11-19);
    the argument marshalling follows what the .pyx does (float32 scores, int32
    ground truth, zero-initialised outputs)."""
    if len(score_matrix_) != len(test_items):
        raise ValueError("The lengths of score_matrix and test_items are not equal.")
    thread_num = thread_num or (os.cpu_count() or 1) * 5
    scores = np.ascontiguousarray(score_matrix_, dtype=np.float32)
    n_users, n_cols = scores.shape
    rankings = np.zeros((n_users, top_k), np.int32)
    lib.ref_top_k_array_index(scores, n_cols, n_users, top_k, thread_num, rankings)
    gts = [np.ascontiguousarray(t, dtype=np.int32) for t in test_items]
    ptrs = (ctypes.POINTER(ctypes.c_int) * n_users)(
        *[t.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) for t in gts])
    lens = np.asarray([len(t) for t in gts], np.int32)
    results = np.zeros((n_users, 5 * top_k), np.float32)
    lib.ref_evaluate_foldout(n_users, rankings, top_k, ptrs, lens, thread_num, results)
    return results, rankings


def run_lgcn(dataset, out):
    install_common_shims()
    lib = load_ref_lib()
    work = tempfile.mkdtemp(prefix="golden_lgcn_")
    if dataset == "tiny":
        write_tiny_dataset(work)
    else:                                         # get_adj_mat writes .npz into the data dir
        shutil.copytree(os.path.join(REF, "data", dataset), os.path.join(work, dataset))
    os.chdir(work)
    sys.path.insert(0, os.path.join(REF, "macr_lightgcn"))
    ev = types.ModuleType("evaluator")
    ev.eval_score_matrix_foldout = lambda s, t, k=20, thread_num=None: \
        ref_eval_score_matrix_foldout(lib, s, t, k, thread_num)[0]
    sys.modules["evaluator"] = ev
    bs = 1024 if dataset != "tiny" else 16
    sys.argv = ["LightGCN.py", "--data_path", work + "/", "--dataset", dataset,
                "--batch_size", str(bs), "--Ks", "[20]" if dataset != "tiny" else "[5, 20]",
                "--layer_size", "[64,64]"]
    import random
    import utility.batch_test as ref_bt          # noqa: E402  (reference module)
    dg = ref_bt.data_generator
    g = {}

    # ---- G1
    g["G1"] = dict(
        n_users=dg.n_users, n_items=dg.n_items, n_train=dg.n_train, n_test=dg.n_test,
        n_exist_users=len(dg.exist_users), exist_users_sha=sha_arr(np.asarray(dg.exist_users, np.int64)),
        train_sha=sha_ints(dg.train_items), test_sha=sha_ints(dg.test_set),
        test_users_order_sha=sha_arr(np.asarray(list(dg.test_set.keys()), np.int64)),
        R_nnz=int(dg.R.nnz))

    # ---- G3 sampler streams (LightGCN.py:651-654 seeds python random + numpy)
    random.seed(SEED)
    np.random.seed(SEED)
    b = np.asarray([dg.sample() for _ in range(3)], dtype=np.int32)
    g3 = dict(batch_size=bs, seed=SEED, sample_sha=sha_arr(b))
    arrays = dict(sample=b)
    if sys.version_info < (3, 11):
        # sample_test draws from dict keys: random.sample(dict_keys) is an error on >=3.11
        random.seed(SEED)
        np.random.seed(SEED)
        if bs <= len(dg.test_set):
            bt = np.asarray([dg.sample_test() for _ in range(2)], dtype=np.int32)
            arrays["sample_test"] = bt
            g3["sample_test_sha"] = sha_arr(bt)
    np.savez_compressed(os.path.join(HERE, "G3_lgcn_sampler_%s.npz" % dataset), **arrays)
    g["G3"] = g3

    # ---- G4 adjacency
    plain, norm, mean, pre = dg.get_adj_mat()
    g4 = {}
    for name, m in (("plain", plain), ("norm", norm), ("mean", mean), ("pre", pre)):
        m = m.tocsr()
        m.sort_indices()
        g4[name] = dict(shape=list(m.shape), nnz=int(m.nnz), dtype=str(m.dtype),
                        indptr_sha=sha_arr(m.indptr.astype(np.int64)),
                        indices_sha=sha_arr(m.indices.astype(np.int64)),
                        data_sha=sha_arr(m.data.astype(np.float32)),
                        data_sum=float(m.data.astype(np.float64).sum()))
    pre = pre.tocsr()
    pre.sort_indices()
    rows = [0, 1, dg.n_users - 1, dg.n_users, dg.n_users + dg.n_items - 1]
    rowdump = {}
    for r in rows:
        s, e = pre.indptr[r], pre.indptr[r + 1]
        rowdump["row%d_idx" % r] = pre.indices[s:e].astype(np.int32)
        rowdump["row%d_val" % r] = pre.data[s:e].astype(np.float32)
    asym = abs(pre - pre.T)
    g4["pre_symmetric_maxabs"] = float(asym.max()) if asym.nnz else 0.0
    g4["rows"] = rows
    np.savez_compressed(os.path.join(HERE, "G4_pre_rows_%s.npz" % dataset), **rowdump)
    if dataset == "tiny":
        np.savez_compressed(os.path.join(HERE, "G4_pre_full_tiny.npz"), indptr=pre.indptr.astype(np.int32),
                            indices=pre.indices.astype(np.int32), data=pre.data.astype(np.float32))
    g["G4"] = g4

    # ---- G6 LightGCN test() via stub session
    users_to_test = list(dg.test_set.keys())
    model = StubModel()
    model.Ks = eval(ref_bt.args.Ks)
    g6 = {}
    kinds = ["normal", "ties", "popular"]
    for kind in kinds:
        full = score_matrix(kind, dg.n_users, dg.n_items, seed=100 + kinds.index(kind))
        for method in ("normal", "rubiboth"):
            sess = StubSession(full)
            ret = ref_bt.test(sess, model, users_to_test, method=method)
            g6["%s/%s" % (kind, method)] = {k: [float(x) for x in v] for k, v in ret.items()}
    g["G6"] = dict(Ks=model.Ks, batch_size=bs, seeds={k: 100 + i for i, k in enumerate(kinds)},
                   results=g6)
    if os.path.exists("Lightgcn_macr.txt"):
        os.remove("Lightgcn_macr.txt")

    # ---- G7 raw C++ evaluator outputs on small stored matrices (inputs + outputs kept)
    if dataset == "tiny":
        rs = np.random.RandomState(77)
        cases = {}
        for name, (u, n, k) in dict(a=(6, 50, 20), b=(5, 12, 10), c=(4, 200, 5), d=(3, 25, 20)).items():
            s = rs.standard_normal((u, n)).astype(np.float32)
            gt = [sorted(rs.choice(n, size=int(rs.randint(1, 6)), replace=False).tolist()) for _ in range(u)]
            if name == "d":                       # K > #unmasked: all but 4 columns at -inf
                s[0, 4:] = -np.inf
                gt[0] = [1, 3]
            res, rank = ref_eval_score_matrix_foldout(lib, s, gt, k, thread_num=4)
            cases[name + "_scores"] = s
            cases[name + "_gt_flat"] = np.asarray([x for t in gt for x in t], np.int32)
            cases[name + "_gt_len"] = np.asarray([len(t) for t in gt], np.int32)
            cases[name + "_k"] = np.asarray(k, np.int32)
            cases[name + "_results"] = res
            cases[name + "_rankings"] = rank
        np.savez_compressed(os.path.join(HERE, "G7_cpp_eval_cases.npz"), **cases)
    import utility.parser as ref_parser            # G9: macr_lightgcn/utility/parser.py:10-104
    argv, sys.argv = sys.argv, ["LightGCN.py"]
    g["G9"] = {k: v for k, v in vars(ref_parser.parse_args()).items()}
    sys.argv = argv
    with open(out, "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    shutil.rmtree(work, ignore_errors=True)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] in ("mf", "lgcn"):
        half, dataset, out = sys.argv[1:4]
        (run_mf if half == "mf" else run_lgcn)(dataset, out)
        return
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"])
    for half in ("mf", "lgcn"):
        for dataset in ("addressa", "tiny"):
            out = os.path.join(HERE, "golden_%s_%s.json" % (half, dataset))
            print("== generating", out, flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), half, dataset, out])
    # the datasets themselves are data fixtures (the reference ships addressa; tiny is ours)
    dst = os.path.join(REPO, "data", "addressa")
    os.makedirs(dst, exist_ok=True)
    for name in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(REF, "data", "addressa", name), os.path.join(dst, name))


if __name__ == "__main__":
    main()
