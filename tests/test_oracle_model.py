"""The model-step half of the oracle (losses, gradients, Adam) checked against
autograd of a LITERAL re-expression of the reference TF graph.

The reference graph cannot be executed here (TensorFlow 1.14 absent), so the
`literal_*` functions below re-type it op for op in torch -- including the
(B,)*(B,1) -> (B,B) broadcast of macr_mf/model.py:204-205 -- and autograd
provides gradients that are independent of the hand derivation in
oracle/macr_oracle.c.
"""
import numpy as np
import pytest
import torch

import oracle


@pytest.fixture(autouse=True, scope="module")
def _float64_autograd():
    """the autograd cross-checks of this module run in float64; the default goes back to what it was afterwards (a
    module-level set_default_dtype leaks into every test collected with this file)"""
    before = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(before)


def literal_normalbce(users, pos_items, neg_items, decay, batch_size):
    # macr_mf/model.py:277-287
    pos_scores = torch.sum(users * pos_items, dim=1)
    neg_scores = torch.sum(users * neg_items, dim=1)
    mf_loss = torch.mean(-torch.log(torch.sigmoid(pos_scores) + 1e-9)
                         + -torch.log(1 - torch.sigmoid(neg_scores) + 1e-9))
    regularizer = (users ** 2).sum() / 2 + (pos_items ** 2).sum() / 2 + (neg_items ** 2).sum() / 2
    regularizer = regularizer / batch_size
    return mf_loss, decay * regularizer


def literal_rubibceboth(users, pos_items, neg_items, w, w_user, alpha, beta, decay, batch_size,
                        reg_rows=None):
    # macr_mf/model.py:185-222  (w, w_user are (d,1) like the TF variables :59-60)
    pos_scores = torch.sum(users * pos_items, dim=1)           # (B,)
    neg_scores = torch.sum(users * neg_items, dim=1)
    pos_item_scores = pos_items @ w                            # (B,1)
    neg_item_scores = neg_items @ w
    user_scores = users @ w_user
    pos_scores = pos_scores * torch.sigmoid(pos_item_scores) * torch.sigmoid(user_scores)   # (B,B)!
    neg_scores = neg_scores * torch.sigmoid(neg_item_scores) * torch.sigmoid(user_scores)
    assert pos_scores.shape == (users.shape[0], users.shape[0])
    mf_loss_ori = torch.mean(-torch.log(torch.sigmoid(pos_scores) + 1e-10)
                             + -torch.log(1 - torch.sigmoid(neg_scores) + 1e-10))
    mf_loss_item = torch.mean(-torch.log(torch.sigmoid(pos_item_scores) + 1e-10)
                              + -torch.log(1 - torch.sigmoid(neg_item_scores) + 1e-10))
    mf_loss_user = torch.mean(-torch.log(torch.sigmoid(user_scores) + 1e-10)
                              + -torch.log(1 - torch.sigmoid(user_scores) + 1e-10))
    mf_loss = mf_loss_ori + alpha * mf_loss_item + beta * mf_loss_user
    ru, ri, rj = reg_rows if reg_rows is not None else (users, pos_items, neg_items)
    regularizer = (ru ** 2).sum() / 2 + (ri ** 2).sum() / 2 + (rj ** 2).sum() / 2
    regularizer = regularizer / batch_size
    return mf_loss, decay * regularizer, (mf_loss_ori, mf_loss_item, mf_loss_user)


def literal_adam(theta, m, v, g, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    # tf.train.AdamOptimizer, TF 1.14
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def make_problem(seed, n_users, n_items, d, B, scale=0.3, dup=True):
    rs = np.random.RandomState(seed)
    P = (rs.standard_normal((n_users, d)) * scale).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * scale).astype(np.float32)
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
    i = rs.randint(0, n_items, B).astype(np.int32)
    j = rs.randint(0, n_items, B).astype(np.int32)
    if dup:                         # hot item, as in Addressa (item 0 in half the lists)
        i[: B // 3] = 0
    return P, Q, w, wu, u, i, j


def autograd_step(kind, P, Q, w, wu, u, i, j, alpha, beta, decay, bs):
    Pt, Qt = torch.tensor(P, dtype=torch.float64, requires_grad=True), torch.tensor(Q, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64).reshape(-1, 1).requires_grad_()
    wut = torch.tensor(wu, dtype=torch.float64).reshape(-1, 1).requires_grad_()
    ul, il, jl = (torch.tensor(x, dtype=torch.long) for x in (u, i, j))
    eu, ei, ej = Pt[ul], Qt[il], Qt[jl]
    if kind == oracle.LOSS_NORMALBCE:
        mf, reg = literal_normalbce(eu, ei, ej, decay, bs)
    else:
        mf, reg, _ = literal_rubibceboth(eu, ei, ej, wt, wut, alpha, beta, decay, bs)
    (mf + reg).backward()
    gw = wt.grad.numpy().ravel() if wt.grad is not None else np.zeros_like(w, dtype=np.float64)
    gwu = wut.grad.numpy().ravel() if wut.grad is not None else np.zeros_like(wu, dtype=np.float64)
    return float(mf.detach()), float(reg.detach()), Pt.grad.numpy(), Qt.grad.numpy(), gw, gwu


@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
@pytest.mark.parametrize("B,d", [(8, 4), (96, 64), (257, 64), (64, 128)])
def test_pair_loss_grad_matches_autograd(kind, B, d):
    P, Q, w, wu, u, i, j = make_problem(B * 7 + d, 300, 50, d, B)
    alpha, beta, decay, bs = 1e-2, 1e-3, 1e-5, 1024
    mf, reg, gP, gQ, gw, gwu = autograd_step(kind, P, Q, w, wu, u, i, j, alpha, beta, decay, bs)
    out = oracle.pair_loss_grad(kind, P[u], Q[i], Q[j], w, wu, alpha, beta)
    assert out["mf"] == pytest.approx(mf, rel=2e-6)
    assert oracle.l2_reg(P[u], Q[i], Q[j], decay, bs) == pytest.approx(reg, rel=2e-6)
    # dense gradients = scatter-add of the pair rows + regulariser rows
    coef = np.float32(decay) / np.float32(bs)
    oP, oQ = np.zeros_like(P, dtype=np.float64), np.zeros_like(Q, dtype=np.float64)
    np.add.at(oP, u, out["deu"] + coef * P[u])
    np.add.at(oQ, i, out["dei"] + coef * Q[i])
    np.add.at(oQ, j, out["dej"] + coef * Q[j])
    scale = max(np.abs(gP).max(), np.abs(gQ).max())
    np.testing.assert_allclose(oP, gP, rtol=2e-4, atol=2e-6 * scale)
    np.testing.assert_allclose(oQ, gQ, rtol=2e-4, atol=2e-6 * scale)
    if kind == oracle.LOSS_RUBIBCEBOTH:
        np.testing.assert_allclose(out["dw"], gw, rtol=2e-4, atol=2e-6 * np.abs(gw).max())
        np.testing.assert_allclose(out["dwu"], gwu, rtol=2e-4, atol=2e-6 * np.abs(gwu).max())


def test_rubibceboth_is_b_by_b_not_per_pair():
    """Pins finding 3 of SURVEY.md: the ori term averages over B*B broadcast entries."""
    P, Q, w, wu, u, i, j = make_problem(3, 40, 30, 8, 16, scale=1.0, dup=False)
    out = oracle.pair_loss_grad(oracle.LOSS_RUBIBCEBOTH, P[u], Q[i], Q[j], w, wu, 0.0, 0.0)
    p, n, si, sj, su = out["fwd"].astype(np.float64)
    sig = lambda x: 1 / (1 + np.exp(-x))
    a, b = sig(si) * sig(su), sig(sj) * sig(su)
    bxb = np.mean(-np.log(sig(a[:, None] * p[None, :]) + 1e-10) - np.log(1 - sig(b[:, None] * n[None, :]) + 1e-10))
    per_pair = np.mean(-np.log(sig(a * p) + 1e-10) - np.log(1 - sig(b * n) + 1e-10))
    assert out["l_ori"] == pytest.approx(bxb, rel=1e-6)
    assert abs(bxb - per_pair) > 1e-3


@pytest.mark.parametrize("kind", [oracle.LOSS_NORMALBCE, oracle.LOSS_RUBIBCEBOTH])
def test_mf_train_steps_match_literal_adam(kind):
    n_users, n_items, d, B = 120, 40, 16, 64
    P, Q, w, wu, _, _, _ = make_problem(11, n_users, n_items, d, B)
    alpha, beta, decay, bs, lr = 1e-2, 1e-3, 1e-5, B, 1e-3
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    ref = dict(P=P.astype(np.float64), Q=Q.astype(np.float64), w=w.astype(np.float64), wu=wu.astype(np.float64))
    slots = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in ref.items()}
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    rs = np.random.RandomState(5)
    for t in range(1, 6):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = rs.randint(0, n_items, B).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        mf, reg, gP, gQ, gw, gwu = autograd_step(kind, ref["P"], ref["Q"], ref["w"], ref["wu"], u, i, j,
                                                 alpha, beta, decay, bs)
        grads = dict(P=gP, Q=gQ, w=gw, wu=gwu)
        upd = ("P", "Q") if kind == oracle.LOSS_NORMALBCE else ("P", "Q", "w", "wu")
        for k in upd:
            ref[k], m, v = literal_adam(ref[k], slots[k][0], slots[k][1], grads[k], t, lr)
            slots[k] = (m, v)
        losses = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        assert losses[1] == pytest.approx(mf, rel=1e-5)
        assert losses[2] == pytest.approx(reg, rel=1e-5)
        assert losses[0] == pytest.approx(mf + reg, rel=1e-5)
        # every row moves every step (dense Adam), so compare whole tables
        np.testing.assert_allclose(Po, ref["P"], rtol=0, atol=3e-6 * t)
        np.testing.assert_allclose(Qo, ref["Q"], rtol=0, atol=3e-6 * t)
        np.testing.assert_allclose(wo, ref["w"], rtol=0, atol=3e-6 * t)
        np.testing.assert_allclose(wuo, ref["wu"], rtol=0, atol=3e-6 * t)
    if kind == oracle.LOSS_NORMALBCE:        # w, w_user receive no gradient -> untouched
        assert np.array_equal(wo, w) and np.array_equal(wuo, wu)
    assert st.power[0] == pytest.approx(0.9 ** 6, rel=1e-6)


def test_oracle_adam_known_answers_from_1e_12_to_1e_1():
    """orc_adam_dense against tf.train.AdamOptimizer's documented rule evaluated in float64 -- lr_t = lr sqrt(1-beta2^t) /
    (1-beta1^t), theta -= lr_t m / (sqrt(v) + epsilon): epsilon outside the root -- over three steps of gradients of every
    magnitude from 1e-12 to 1e-1 on tables that start at zero (the value IS the sum of the updates).  Around |g| ~ 3e-7,
    where sqrt(v) ~ epsilon, the other forms of the rule are off by factors; the -m gpu test of the same name holds the HIP
    pass to the same answers."""
    from oracle.oracle import _ptr
    lr, b1, b2, eps = (float(np.float32(x)) for x in (1e-3, 0.9, 0.999, 1e-8))      # as the fp32 graph holds them
    rs = np.random.RandomState(5)
    n = 4096
    theta, m, v = (np.zeros(n, np.float32) for _ in range(3))
    power = np.array([b1, b2], np.float32)
    t64, m64, v64 = (np.zeros(n) for _ in range(3))
    wrong = np.zeros(n)                                  # epsilon inside the root
    m_tol = np.zeros(n)
    for t in range(1, 4):
        g = (10.0 ** rs.uniform(-12, -1, n) * rs.choice([-1.0, 1.0], n)).astype(np.float32)
        lr_t = oracle.lib().orc_adam_lr_t(lr, power)
        oracle.lib().orc_adam_dense(theta, m, v, _ptr(g), n, lr_t, b1, b2, eps)
        p1, p2 = float(power[0]), float(power[1])
        assert lr_t == pytest.approx(lr * np.sqrt(1 - p2) / (1 - p1), rel=1e-6)
        m_terms = np.abs(b1 * m64) + np.abs((1 - b1) * g.astype(np.float64))        # (gradients of either sign: m may cancel)
        m64 = b1 * m64 + (1 - b1) * g.astype(np.float64)
        v64 = b2 * v64 + (1 - b2) * g.astype(np.float64) ** 2
        t64 = t64 - lr * np.sqrt(1 - p2) / (1 - p1) * m64 / (np.sqrt(v64) + eps)
        wrong = wrong - lr * np.sqrt(1 - p2) / (1 - p1) * m64 / np.sqrt(v64 + eps)
        power *= np.array([b1, b2], np.float32)
        m_tol = 2.5e-7 * m_terms + b1 * m_tol            # (three roundings per step on the terms, and what was inherited)
        assert np.all(np.abs(m - m64) <= m_tol)
        np.testing.assert_allclose(v, v64, rtol=2e-6, atol=1e-44)
        # (sums of three updates of either sign: a relative bound on the value, an absolute one where they cancel)
        np.testing.assert_allclose(theta, t64, rtol=1e-5, atol=1e-9)
    assert np.max(np.abs(wrong - theta) / np.maximum(np.abs(theta), 1e-12)) > 0.1


def _toy_graph(n_users, n_items, seed):
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    R = (rs.rand(n_users, n_items) < 0.15).astype(np.float32)
    R[0, :] = 0                      # an isolated user: zero degree row
    A = sp.bmat([[None, sp.csr_matrix(R)], [sp.csr_matrix(R.T), None]]).tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5)
    dinv[np.isinf(dinv)] = 0
    A_hat = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A_hat.sort_indices()
    return A_hat


def test_lgcn_propagate_and_train_step_match_autograd():
    n_users, n_items, d, B, L = 30, 20, 8, 24, 2
    A = _toy_graph(n_users, n_items, 1)
    P, Q, w, wu, u, i, j = make_problem(2, n_users, n_items, d, B, dup=False)
    T = np.concatenate([P, Q]).astype(np.float32)
    E = oracle.lgcn_propagate(A.indptr, A.indices, A.data, T, L)
    Ad = A.toarray().astype(np.float64)
    T64 = T.astype(np.float64)
    want = (T64 + Ad @ T64 + Ad @ Ad @ T64) / 3            # LightGCN.py:297-307
    np.testing.assert_allclose(E, want, rtol=1e-5, atol=1e-6)

    alpha, beta, decay, bs, lr = 1e-2, 1e-3, 1e-4, B, 1e-2
    kind = oracle.LOSS_RUBIBCEBOTH
    Tt = torch.tensor(T64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64).reshape(-1, 1).requires_grad_()
    wut = torch.tensor(wu, dtype=torch.float64).reshape(-1, 1).requires_grad_()
    At = torch.tensor(Ad)
    Et = (Tt + At @ Tt + At @ (At @ Tt)) / 3
    ul, il, jl = (torch.tensor(x, dtype=torch.long) for x in (u, i, j))
    mf, reg, _ = literal_rubibceboth(Et[ul], Et[n_users + il], Et[n_users + jl], wt, wut, alpha, beta, decay, bs,
                                     reg_rows=(Tt[ul], Tt[n_users + il], Tt[n_users + jl]))   # LightGCN.py:525-527
    (mf + reg).backward()
    Tn, _, _ = literal_adam(T64, 0, 0, Tt.grad.numpy(), 1, lr)
    wn, _, _ = literal_adam(w.astype(np.float64), 0, 0, wt.grad.numpy().ravel(), 1, lr)
    st = oracle.AdamState([T.shape, (d,), (d,)])
    To, wo, wuo = T.copy(), w.copy(), wu.copy()
    losses = oracle.lgcn_train_step(kind, n_users, n_items, L, A.indptr, A.indices, A.data, u, i, j,
                                    To, wo, wuo, st, lr, decay, alpha, beta, bs)
    assert losses[1] == pytest.approx(float(mf), rel=1e-5)
    assert losses[2] == pytest.approx(float(reg), rel=1e-5)
    # Adam's first step is lr*sign(g) wherever |g| >> eps: compare where the gradient is not tiny
    g = Tt.grad.numpy()
    big = np.abs(g) > 1e-6
    np.testing.assert_allclose(To[big], Tn[big], rtol=0, atol=2e-5)
    np.testing.assert_allclose(wo, wn, rtol=0, atol=2e-5)
