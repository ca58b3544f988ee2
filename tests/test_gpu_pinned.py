"""HIP training steps against G10 (-m gpu): losses and gradients computed by the reference's own loss-graph code
(tests/golden/make_golden_model.py), through the C ABI.  Gradients are read off the first Adam step (m = 0.1 * g)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G10 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G10_model_steps.npz"))
ALPHA, BETA, DECAY, BS = (float(G10["hyper"][0]), float(G10["hyper"][1]), float(G10["hyper"][2]), int(G10["hyper"][3]))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from macr_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def close_grad(got, want, name, rtol):
    np.testing.assert_allclose(got, want, rtol=rtol, atol=2e-6 * np.abs(want).max() + 1e-12, err_msg=name)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("loss", ["normalbce", "rubibceboth", "rubibce"])
def test_hip_mf_step_matches_reference_graph(ops, tag, loss):
    kind = {"normalbce": ops.LOSS_NORMALBCE, "rubibceboth": ops.LOSS_RUBIBCEBOTH, "rubibce": ops.LOSS_RUBIBCE}[loss]
    g = lambda k: G10["mf_%s/%s" % (tag, k)]
    state = ops.MFState(dev(g("P")), dev(g("Q")), dev(g("w").reshape(-1)), dev(g("wu").reshape(-1)),
                        ops.make_hyper(1e-3, DECAY, ALPHA, BETA, BS), len(g("u")))
    got = state.step(kind, dev(g("u"), torch.int32), dev(g("i"), torch.int32), dev(g("j"), torch.int32)).cpu().numpy()
    # case c saturates fp32 (1 - sigmoid(x) == 0 beyond x ~ 17): only the fp32 execution of the graph is the truth there
    for dt, rtol in (("f32", 1e-5),) + ((("f64", 1e-5),) if tag != "c" else ()):
        want = [float(g("%s/%s/%s" % (loss, dt, k))) for k in ("loss", "mf_loss", "reg_loss")]
        np.testing.assert_allclose(got, want, rtol=rtol, err_msg=dt)
    pre = "%s/f32/" % loss
    close_grad(state.mP.cpu().numpy() / 0.1, g(pre + "dP"), "dP", 2e-4)
    close_grad(state.mQ.cpu().numpy() / 0.1, g(pre + "dQ"), "dQ", 2e-4)
    if loss != "normalbce":
        close_grad(state.mw.cpu().numpy() / 0.1, g(pre + "dw").reshape(-1), "dw", 2e-4)
    if loss == "rubibceboth":
        close_grad(state.mwu.cpu().numpy() / 0.1, g(pre + "dwu").reshape(-1), "dwu", 2e-4)
    else:                      # w_user gets no gradient outside rubibceboth: value and slots bitwise untouched
        assert np.array_equal(state.wu.cpu().numpy(), g("wu").reshape(-1))
        assert not state.mwu.cpu().numpy().any() and not state.vwu.cpu().numpy().any()


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("loss", ["bce", "bceboth"])
def test_hip_lightgcn_step_matches_reference_graph(ops, tag, loss):
    kind = ops.LOSS_NORMALBCE if loss == "bce" else ops.LOSS_RUBIBCEBOTH
    g = lambda k: G10["lgcn_%s/%s" % (tag, k)]
    P, Q = g("P"), g("Q")
    adj = ops.CSR(dev(g("indptr"), torch.int32), dev(g("indices"), torch.int32), dev(g("data"))).build_spmm_plan()
    state = ops.LGCNState(dev(np.concatenate([P, Q])), P.shape[0], Q.shape[0], dev(g("w").reshape(-1)),
                          dev(g("wu").reshape(-1)), adj, 2, ops.make_hyper(1e-3, DECAY, ALPHA, BETA, BS), len(g("u")))
    E = state.propagated().cpu().numpy()
    np.testing.assert_allclose(E, np.concatenate([g("bce/f64/ua"), g("bce/f64/ia")]), rtol=2e-5, atol=1e-7)
    got = state.step(kind, dev(g("u"), torch.int32), dev(g("i"), torch.int32), dev(g("j"), torch.int32)).cpu().numpy()
    for dt in ("f32", "f64"):
        want = [float(g("%s/%s/%s" % (loss, dt, k))) for k in ("loss", "mf_loss", "emb_loss")]
        np.testing.assert_allclose(got, want, rtol=1e-5, err_msg=dt)
    pre = "%s/f64/" % loss
    close_grad(state.mT.cpu().numpy() / 0.1, np.concatenate([g(pre + "dP"), g(pre + "dQ")]), "dT", 5e-4)
    if kind == ops.LOSS_RUBIBCEBOTH:
        close_grad(state.mw.cpu().numpy() / 0.1, g(pre + "dw").reshape(-1), "dw", 5e-4)
        close_grad(state.mwu.cpu().numpy() / 0.1, g(pre + "dwu").reshape(-1), "dwu", 5e-4)


@pytest.mark.parametrize("c", [0.0, 40.0])
def test_hip_score_kinds_match_reference_graph(ops, c):
    """model.py:45, :141-142, :199-201 for all users x all items (64 x 64, d = 32): the dense scores against the
    reference graph's tensors, and bit for bit against the oracle (same fma chain, same epilogue roundings)."""
    import oracle
    P, Q, w, wu = (G10["mf_scores/%s" % k] for k in ("P", "Q", "w", "wu"))
    Pd, Qd = dev(P), dev(Q)
    sig_i = ops.branch_sigmoid(Qd, dev(w.reshape(-1)))
    sig_u = ops.branch_sigmoid(Pd, dev(wu.reshape(-1)))
    for name, kind in (("batch_ratings", ops.SCORE_NORMAL), ("rubi_ratings_both", ops.SCORE_RUBI_BOTH),
                       ("rubi_ratings", ops.SCORE_RUBI), ("direct_minus_ratings", ops.SCORE_DIRECT_MINUS),
                       ("direct_minus_ratings_both", ops.SCORE_DIRECT_MINUS_BOTH)):
        got = ops.score_matrix(kind, Pd, None, Qd, sig_u, sig_i, c).cpu().numpy()
        want = G10["mf_scores/c%g/f64/%s" % (c, name)]
        np.testing.assert_allclose(got, want, rtol=3e-6, atol=3e-6 * np.abs(want).max(), err_msg=name)
        orc = oracle.score_matrix(kind, P, Q, sig_u.cpu().numpy(), sig_i.cpu().numpy(), c)
        assert np.array_equal(got.view(np.uint32), orc.view(np.uint32)), name


@pytest.mark.parametrize("kind_name", ["SCORE_RUBI", "SCORE_DIRECT_MINUS", "SCORE_DIRECT_MINUS_BOTH"])
def test_hip_topk_of_the_other_score_kinds_is_bit_exact(ops, kind_name):
    import oracle
    kind = getattr(ops, kind_name)
    rs = np.random.RandomState(17)
    U, N, d, K = 700, 5000, 64, 20
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    mask = [sorted(rs.choice(N, 30, replace=False).tolist()) for _ in range(U)]
    sig_i = ops.branch_sigmoid(dev(Q), dev(w))
    sig_u = ops.branch_sigmoid(dev(P), dev(wu))
    v, ix = ops.score_topk(kind, dev(P), None, dev(Q), K, sig_u, sig_i, 25.0, ops.CSR.from_lists(mask, "cuda"))
    val, idx, cnt = ops.topk_merge(v, ix)
    wv, wi, wc = oracle.score_topk(kind, P, Q, K, sig_u.cpu().numpy(), sig_i.cpu().numpy(), 25.0, oracle.csr_from_lists(mask))
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(val.cpu().numpy().view(np.uint32), wv.view(np.uint32))
