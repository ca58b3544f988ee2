"""The model-step half of the oracle against G10: losses and gradients computed by the REFERENCE's own loss-graph
code (tests/golden/make_golden_model.py runs macr_mf/model.py and macr_lightgcn/LightGCN.py builders through a
functional tensorflow stand-in).  Pins oracle/macr_oracle.c's restatement of model.py:185-222 / :277-287 and
LightGCN.py:288-309 / :415-429 / :495-532; tf.train.AdamOptimizer itself stays unpinned (TF absent), so gradients are
read off the first Adam step, where m = (1 - beta1) * g exactly."""
import os

import numpy as np
import pytest

import oracle

G10 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G10_model_steps.npz"))
ALPHA, BETA, DECAY, BS = (float(G10["hyper"][0]), float(G10["hyper"][1]), float(G10["hyper"][2]), int(G10["hyper"][3]))
KINDS = {"normalbce": oracle.LOSS_NORMALBCE, "rubibceboth": oracle.LOSS_RUBIBCEBOTH, "rubibce": oracle.LOSS_RUBIBCE}


def g(key):
    return G10[key]


def close_grad(got, want, name, rtol=2e-4):
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=2e-6 * scale + 1e-12, err_msg=name)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("loss", sorted(KINDS))
def test_oracle_mf_step_matches_reference_graph(tag, loss):
    P, Q, w, wu = (g("mf_%s/%s" % (tag, k)).copy() for k in ("P", "Q", "w", "wu"))
    u, i, j = (g("mf_%s/%s" % (tag, k)).astype(np.int32) for k in ("u", "i", "j"))
    w, wu = w.reshape(-1).copy(), wu.reshape(-1).copy()
    st = oracle.AdamState([P.shape, Q.shape, w.shape, wu.shape])
    got = oracle.mf_train_step(KINDS[loss], u, i, j, P, Q, w, wu, st, 1e-3, DECAY, ALPHA, BETA, BS)
    # Case c has logits beyond +-17, where fp32 `1 - sigmoid(x)` is exactly 0 and the loss term becomes log(1e-9):
    # there the fp32 execution of the reference's graph is the truth (the fp64 run differs by 0.9 %), and the oracle
    # must reproduce the saturation.  Cases a, b are also checked against the fp64 execution.
    for dt, rtol in (("f32", 3e-6),) + ((("f64", 1e-5),) if tag != "c" else ()):     # north star: loss within 1e-5 relative
        want = [float(g("mf_%s/%s/%s/%s" % (tag, loss, dt, k))) for k in ("loss", "mf_loss", "reg_loss")]
        np.testing.assert_allclose(got, want, rtol=rtol, err_msg=dt)
    for dt, rtol in (("f32", 5e-5),) + ((("f64", 2e-4),) if tag != "c" else ()):
        pre = "mf_%s/%s/%s/" % (tag, loss, dt)
        close_grad(st.m[0] / 0.1, g(pre + "dP"), "dP " + dt, rtol)
        close_grad(st.m[1] / 0.1, g(pre + "dQ"), "dQ " + dt, rtol)
        if loss != "normalbce":
            close_grad(st.m[2] / 0.1, g(pre + "dw").reshape(-1), "dw " + dt, rtol)
        else:                                                  # normalbce: w receives no gradient (model.py:95)
            assert not g(pre + "dw").any() and not st.m[2].any()
        if loss == "rubibceboth":
            close_grad(st.m[3] / 0.1, g(pre + "dwu").reshape(-1), "dwu " + dt, rtol)
        else:                                                  # w_user: only rubibceboth's graph reaches it
            assert not g(pre + "dwu").any() and not st.m[3].any()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_lightgcn_propagation_matches_reference_graph(tag):
    P, Q = g("lgcn_%s/P" % tag), g("lgcn_%s/Q" % tag)
    E = oracle.lgcn_propagate(g("lgcn_%s/indptr" % tag), g("lgcn_%s/indices" % tag), g("lgcn_%s/data" % tag),
                              np.concatenate([P, Q]), 2)
    want = np.concatenate([g("lgcn_%s/bce/f64/ua" % tag), g("lgcn_%s/bce/f64/ia" % tag)])
    np.testing.assert_allclose(E, want, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("loss,kind", [("bce", oracle.LOSS_NORMALBCE), ("bceboth", oracle.LOSS_RUBIBCEBOTH)])
def test_oracle_lightgcn_step_matches_reference_graph(tag, loss, kind):
    P, Q, w, wu = (g("lgcn_%s/%s" % (tag, k)).copy() for k in ("P", "Q", "w", "wu"))
    u, i, j = (g("lgcn_%s/%s" % (tag, k)).astype(np.int32) for k in ("u", "i", "j"))
    T = np.concatenate([P, Q])
    w, wu = w.reshape(-1).copy(), wu.reshape(-1).copy()
    st = oracle.AdamState([T.shape, w.shape, wu.shape])
    got = oracle.lgcn_train_step(kind, P.shape[0], Q.shape[0], 2, g("lgcn_%s/indptr" % tag), g("lgcn_%s/indices" % tag),
                                 g("lgcn_%s/data" % tag), u, i, j, T, w, wu, st, 1e-3, DECAY, ALPHA, BETA, BS)
    for dt, rtol in (("f32", 5e-6), ("f64", 1e-5)):
        want = [float(g("lgcn_%s/%s/%s/%s" % (tag, loss, dt, k))) for k in ("loss", "mf_loss", "emb_loss")]
        np.testing.assert_allclose(got, want, rtol=rtol, err_msg=dt)
    pre = "lgcn_%s/%s/f64/" % (tag, loss)
    close_grad(st.m[0] / 0.1, np.concatenate([g(pre + "dP"), g(pre + "dQ")]), "dT", rtol=3e-4)
    if kind == oracle.LOSS_RUBIBCEBOTH:
        close_grad(st.m[1] / 0.1, g(pre + "dw").reshape(-1), "dw", rtol=3e-4)
        close_grad(st.m[2] / 0.1, g(pre + "dwu").reshape(-1), "dwu", rtol=3e-4)


@pytest.mark.parametrize("c", [0.0, 40.0])
def test_oracle_score_kinds_match_reference_graph(c):
    """Test-time score tensors of model.py:45, :141-142, :199-201 for all users x all items (64 x 64, d = 32)."""
    P, Q, w, wu = (g("mf_scores/%s" % k) for k in ("P", "Q", "w", "wu"))
    sig_i = oracle.branch_sigmoid(Q, w.reshape(-1))
    sig_u = oracle.branch_sigmoid(P, wu.reshape(-1))
    for name, kind in (("batch_ratings", oracle.SCORE_NORMAL), ("rubi_ratings_both", oracle.SCORE_RUBI_BOTH),
                       ("rubi_ratings", oracle.SCORE_RUBI), ("direct_minus_ratings", oracle.SCORE_DIRECT_MINUS),
                       ("direct_minus_ratings_both", oracle.SCORE_DIRECT_MINUS_BOTH)):
        got = oracle.score_matrix(kind, P, Q, sig_u, sig_i, c)
        for dt, rtol in (("f32", 2e-6), ("f64", 2e-6)):
            want = g("mf_scores/c%g/%s/%s" % (c, dt, name))
            np.testing.assert_allclose(got, want, rtol=rtol, atol=2e-6 * np.abs(want).max(), err_msg="%s %s" % (name, dt))
