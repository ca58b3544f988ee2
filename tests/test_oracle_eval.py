"""Evaluator half of the oracle, pinned against
  * G5/G6: the reference's own test() functions (macr_mf/train.py:162, utility/batch_test.py:26) run
           through a stub session on regenerable score matrices,
  * G7:    raw outputs of the reference's C++ evaluator on stored matrices,
  * G8:    the reference's metric functions on hand-made hit vectors,
  * oracle/_ref: the reference C++ evaluator itself, compiled here from /root/reference (when present).
"""
import os

import numpy as np
import pytest

import oracle
from macr_amd import metrics_host
from macr_amd.data import LGCNData, MFData
from helpers import GOLD, dataset_args, golden, masked_scores, score_matrix


def oracle_mf_test(data, users, scores_full, Ks):
    mask, gt = data.eval_lists(users)
    s = scores_full[np.asarray(users)]
    _, idx, cnt = oracle.topk_scores(s, max(Ks), oracle.csr_from_lists(mask))
    per_user = oracle.metrics_mf(idx, cnt, oracle.csr_from_lists(gt), Ks)
    m = per_user.mean(0)
    return dict(precision=m[0], recall=m[1], ndcg=m[2], hit_ratio=m[3])


def oracle_lgcn_test(dg, users, scores_full, Ks):
    mask, gt = dg.eval_lists(users)
    top_show = np.sort(np.asarray(Ks))
    max_top = int(top_show.max())
    s = masked_scores(scores_full[np.asarray(users)], mask)             # batch_test.py:124-129
    _, idx, _ = oracle.topk_scores(s, max_top)
    res = oracle.metrics_foldout(idx, oracle.csr_from_lists(gt))
    res[:, 2 * max_top:3 * max_top] = (res[:, max_top:2 * max_top] != 0)     # :143-149
    final = res.astype(np.float64).mean(0).reshape(5, max_top)[:, top_show - 1]
    return dict(hr=final[2], recall=final[1], ndcg=final[3])


@pytest.mark.parametrize("dataset", ["addressa", "tiny"])
def test_G5_mf_evaluator(dataset):
    g = golden("mf", dataset)["G5"]
    data = MFData(dataset_args(dataset))
    users = list(data.test_user_list.keys())
    for kind, seed in g["seeds"].items():
        full = score_matrix(kind, data.n_users, data.n_items, seed)
        got = oracle_mf_test(data, users, full, g["Ks"])
        for model_type in ("o", "rubi_both"):
            want = g["results"]["%s/%s" % (kind, model_type)]
            for k in ("precision", "recall", "ndcg", "hit_ratio"):
                np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=1e-15, err_msg="%s %s" % (kind, k))


@pytest.mark.parametrize("dataset", ["addressa", "tiny"])
def test_G6_lgcn_evaluator(dataset):
    g = golden("lgcn", dataset)["G6"]
    a = dataset_args(dataset)
    dg = LGCNData(path=a.data_path + a.dataset, batch_size=a.batch_size, args=a)
    users = list(dg.test_set.keys())
    for kind, seed in g["seeds"].items():
        if kind == "ties":
            continue        # std::partial_sort_copy leaves tie order unspecified; our rule is id-ascending
        full = score_matrix(kind, dg.n_users, dg.n_items, seed)
        got = oracle_lgcn_test(dg, users, full, g["Ks"])
        for method in ("normal", "rubiboth"):
            want = g["results"]["%s/%s" % (kind, method)]
            for k in ("hr", "recall", "ndcg"):
                np.testing.assert_allclose(got[k], want[k], rtol=2e-6, err_msg="%s %s" % (kind, k))


def test_G6_ties_differ_only_by_tie_order():
    """On tie-heavy scores the two reference evaluators disagree with EACH OTHER (heapq.nlargest is
    stable, std::partial_sort_copy is not); the oracle follows the MF (stable, id-ascending) rule."""
    mf = golden("mf", "addressa")["G5"]["results"]["ties/o"]
    lg = golden("lgcn", "addressa")["G6"]["results"]["ties/normal"]
    assert abs(mf["hit_ratio"][0] - lg["hr"][0]) > 1e-4
    nm = golden("mf", "addressa")["G5"]["results"]["normal/o"]
    nl = golden("lgcn", "addressa")["G6"]["results"]["normal/normal"]
    assert abs(nm["hit_ratio"][0] - nl["hr"][0]) < 1e-7          # tie-free: they agree


def test_G7_cpp_evaluator_cases():
    z = np.load(os.path.join(GOLD, "G7_cpp_eval_cases.npz"))
    for name in "abcd":
        s, k = z[name + "_scores"], int(z[name + "_k"])
        gptr = np.concatenate([[0], np.cumsum(z[name + "_gt_len"])]).astype(np.int32)
        _, idx, _ = oracle.topk_scores(s, k)
        if name != "d":
            assert np.array_equal(idx, z[name + "_rankings"])
        else:                                   # only 4 finite columns: the finite prefix is pinned
            assert np.array_equal(idx[0, :4], z["d_rankings"][0, :4])
            assert np.array_equal(idx[1:], z["d_rankings"][1:])
        res = oracle.metrics_foldout(idx, (gptr, z[name + "_gt_flat"]))
        np.testing.assert_allclose(res, z[name + "_results"], rtol=1e-7, atol=0)


def test_G8_metric_unit_cases():
    for case in golden("mf", "tiny")["G8"]:
        r, k, npos = case["r"], case["k"], case["n_pos"]
        assert metrics_host.precision_at_k(r, k) == pytest.approx(case["precision"], rel=1e-15)
        assert metrics_host.recall_at_k(r, k, npos) == pytest.approx(case["recall"], rel=1e-15)
        assert metrics_host.ndcg_at_k(r, k, npos) == pytest.approx(case["ndcg"], rel=1e-15)
        assert metrics_host.hit_at_k(r, k) == case["hit"]
        assert metrics_host.dcg_at_k(r, k) == pytest.approx(case["dcg"], rel=1e-15)
        if sum(r[:k]) > npos:
            continue                # hand-made vector with more hits than positives: no ranking realises it
        # and the oracle's per-user kernel on the same hit pattern
        rank = np.asarray([[i if h else 1000 + i for i, h in enumerate(r)] + [-1] * (k - len(r))], np.int32)[:, :k]
        gt_items = sorted([i for i, h in enumerate(r) if h])
        gt_items += list(range(2000, 2000 + npos - len(gt_items)))
        out = oracle.metrics_mf(rank, np.asarray([min(len(r), k)], np.int32), oracle.csr_from_lists([gt_items]), [k])
        np.testing.assert_allclose(out[0, :, 0], [case["precision"], case["recall"], case["ndcg"], case["hit"]],
                                   rtol=1e-12)


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_against_compiled_reference_evaluator():
    rs = np.random.RandomState(12)
    for U, N, K in ((50, 744, 20), (8, 40981, 20), (20, 100, 30)):
        s = rs.standard_normal((U, N)).astype(np.float32)
        gt = [sorted(rs.choice(N, size=rs.randint(1, 12), replace=False).tolist()) for _ in range(U)]
        mask = [sorted(rs.choice(N, size=rs.randint(0, 40), replace=False).tolist()) for _ in range(U)]
        sm = masked_scores(s, mask)
        ref_res, ref_rank = oracle.ref_eval_score_matrix_foldout(sm, gt, K)
        _, idx, _ = oracle.topk_scores(sm, K)
        assert np.array_equal(idx, ref_rank)
        np.testing.assert_allclose(oracle.metrics_foldout(idx, oracle.csr_from_lists(gt)), ref_res, rtol=1e-7, atol=0)
        # masking by candidate removal (MF) ranks the same items as masking by -inf (LightGCN)
        _, idx2, _ = oracle.topk_scores(s, K, oracle.csr_from_lists(mask))
        assert np.array_equal(idx2, ref_rank)
