"""oracle/torch_port.py (the torch-CPU restatement of the reference's graph that bench.py times as `cpu_baseline`) against
the C oracle: autograd must reproduce the hand-derived gradients, AdamTF the oracle's Adam, the evaluation the oracle's
ranking.  Both are test/bench infrastructure; this pins the baseline to the checker so that what is timed is the path."""
import numpy as np
import scipy.sparse as sp
import torch

import oracle
from oracle import torch_port as tp


def _mf_problem(seed=3, n_users=400, n_items=300, d=32, B=128):
    rs = np.random.RandomState(seed)
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    batches = [(rs.choice(n_users, B, replace=False).astype(np.int32), (rs.zipf(1.3, B) % n_items).astype(np.int32),
                rs.randint(0, n_items, B).astype(np.int32)) for _ in range(3)]
    return P, Q, w, wu, batches


def test_mf_port_steps_equal_the_oracle():
    lr, decay, alpha, beta, bs = 1e-3, 1e-5, 1e-2, 1e-3, 96
    for kind in (oracle.LOSS_RUBIBCEBOTH, oracle.LOSS_NORMALBCE):
        P, Q, w, wu, batches = _mf_problem()
        port = tp.MFPort(P, Q, w, wu, lr, decay, alpha, beta, bs)
        Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
        st = oracle.AdamState([P.shape, Q.shape, w.shape, wu.shape])
        for u, i, j in batches:
            got = port.train_step(tp.LOSS_RUBIBCEBOTH if kind == oracle.LOSS_RUBIBCEBOTH else tp.LOSS_NORMALBCE, u, i, j)
            want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
            np.testing.assert_allclose(got, want, rtol=2e-6)
        # three Adam steps move a touched row by ~3e-3: agreement to 1e-6 pins gradients, duplicate sums and the update rule
        np.testing.assert_allclose(port.P.detach().numpy(), Po, atol=2e-6)
        np.testing.assert_allclose(port.Q.detach().numpy(), Qo, atol=2e-6)
        np.testing.assert_allclose(port.w.detach().numpy().ravel(), wo, atol=2e-6)
        np.testing.assert_allclose(port.wu.detach().numpy().ravel(), wuo, atol=2e-6)
        if kind == oracle.LOSS_NORMALBCE:
            assert np.array_equal(port.w.detach().numpy().ravel(), w) and np.array_equal(port.wu.detach().numpy().ravel(), wu)


def test_mf_port_evaluation_equals_the_oracle_ranking():
    P, Q, w, wu, _ = _mf_problem(seed=8)
    rs = np.random.RandomState(1)
    users = np.sort(rs.choice(P.shape[0], 150, replace=False)).astype(np.int32)
    mask = [sorted(rs.choice(Q.shape[0], 12, replace=False).tolist()) for _ in users]
    port = tp.MFPort(P, Q, w, wu, 1e-3, 1e-5, 1e-2, 1e-3, 64)
    got = port.evaluate(users, mask, 40.0, K=20, batch=64)
    sig_i, sig_u = oracle.branch_sigmoid(Q, w), oracle.branch_sigmoid(P[users], wu)
    _, want, _ = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P[users], Q, 20, sig_u, sig_i, 40.0, oracle.csr_from_lists(mask))
    assert (got == want).mean() > 0.999        # (torch's matmul sums in another order: a near-tie may swap)


def test_lgcn_port_steps_equal_the_oracle():
    rs = np.random.RandomState(5)
    n_u, n_i, d, B, L = 120, 90, 32, 64, 2
    R = sp.random(n_u, n_i, density=0.08, random_state=rs, format="csr", dtype=np.float32)
    R.data[:] = 1.0
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(1)).ravel()
    dinv = np.where(deg > 0, np.power(np.maximum(deg, 1), -0.5), 0).astype(np.float32)
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A.sort_indices()
    N = n_u + n_i
    T = (rs.standard_normal((N, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    lr, decay, alpha, beta = 1e-3, 1e-4, 1e-2, 1e-3
    for kind in (oracle.LOSS_RUBIBCEBOTH, oracle.LOSS_NORMALBCE):
        port = tp.LGCNPort(T, n_u, n_i, w, wu, tp.csr_to_torch(A.indptr, A.indices, A.data, N), L, lr, decay, alpha, beta, B)
        To, wo, wuo = T.copy(), w.copy(), wu.copy()
        st = oracle.AdamState([T.shape, (d,), (d,)])
        for _ in range(3):
            u = rs.choice(n_u, B, replace=False).astype(np.int32)
            i, j = rs.randint(0, n_i, B).astype(np.int32), rs.randint(0, n_i, B).astype(np.int32)
            got = port.train_step(tp.LOSS_RUBIBCEBOTH if kind == oracle.LOSS_RUBIBCEBOTH else tp.LOSS_NORMALBCE, u, i, j)
            want = oracle.lgcn_train_step(kind, n_u, n_i, L, A.indptr, A.indices, A.data, u, i, j, To, wo, wuo, st, lr, decay,
                                          alpha, beta, B)
            np.testing.assert_allclose(got, want, rtol=3e-6)
        np.testing.assert_allclose(port.T.detach().numpy(), To, atol=3e-6)
        E = tp.lgcn_propagate(port.A, port.T.detach(), L).numpy()
        np.testing.assert_allclose(E, oracle.lgcn_propagate(A.indptr, A.indices, A.data, To, L), atol=3e-6)


def test_lgcn_steps_on_a_row_normalised_adjacency_need_the_transposed_operator():
    """--adj_type norm / gcmc / mean (LightGCN.py:667-678) are D^-1 A: not symmetric.  The gradient of A @ E is A^T @ dE
    (tf.gradients of tf.sparse_tensor_dense_matmul): autograd of the torch port does that by itself, the oracle takes A^T
    explicitly -- they must agree, and the step with A in place of A^T must NOT (the test would be blind otherwise)."""
    rs = np.random.RandomState(11)
    n_u, n_i, d, B, L = 100, 70, 32, 48, 2
    R = sp.random(n_u, n_i, density=0.1, random_state=rs, format="csr", dtype=np.float32)
    R.data[:] = 1.0
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(1)).ravel()
    A = (sp.diags(np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0).astype(np.float32)) @ A).tocsr().astype(np.float32)   # gcmc: D^-1 A
    A.sort_indices()
    AT = A.T.tocsr().astype(np.float32)
    AT.sort_indices()
    assert abs(A - AT).max() > 1e-3
    N = n_u + n_i
    T = (rs.standard_normal((N, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    lr, decay, alpha, beta = 1e-3, 1e-4, 1e-2, 1e-3
    port = tp.LGCNPort(T, n_u, n_i, w, wu, tp.csr_to_torch(A.indptr, A.indices, A.data, N), L, lr, decay, alpha, beta, B)
    To, wo, wuo = T.copy(), w.copy(), wu.copy()
    Tw = T.copy()
    st, stw = oracle.AdamState([T.shape, (d,), (d,)]), oracle.AdamState([T.shape, (d,), (d,)])
    for _ in range(3):
        u = rs.choice(n_u, B, replace=False).astype(np.int32)
        i, j = rs.randint(0, n_i, B).astype(np.int32), rs.randint(0, n_i, B).astype(np.int32)
        got = port.train_step(tp.LOSS_RUBIBCEBOTH, u, i, j)
        want = oracle.lgcn_train_step(oracle.LOSS_RUBIBCEBOTH, n_u, n_i, L, A.indptr, A.indices, A.data, u, i, j, To, wo, wuo, st,
                                      lr, decay, alpha, beta, B, transposed=(AT.indptr, AT.indices, AT.data))
        oracle.lgcn_train_step(oracle.LOSS_RUBIBCEBOTH, n_u, n_i, L, A.indptr, A.indices, A.data, u, i, j, Tw, w.copy(), wu.copy(), stw,
                               lr, decay, alpha, beta, B)
        np.testing.assert_allclose(got, want, rtol=3e-6)
    np.testing.assert_allclose(port.T.detach().numpy(), To, atol=3e-6)
    assert np.abs(Tw - To).max() > 1e-4          # (A instead of A^T: a different model after three steps)


def test_fast_build_of_the_c_port_equals_the_checker():
    """bench.py times oracle/_build/libmacr_oracle_fast.so (same C file, -O3 AVX2 -ffast-math) as the tuned CPU port: it
    must compute the checker's step -- losses to 1e-6 relative, tables to 1e-6 absolute after three steps."""
    lr, decay, alpha, beta, bs = 1e-3, 1e-5, 1e-2, 1e-3, 96
    P, Q, w, wu, batches = _mf_problem(seed=11, n_users=500, n_items=350, d=64, B=256)
    outs = []
    for use_fast in (False, True):
        Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
        st = oracle.AdamState([P.shape, Q.shape, w.shape, wu.shape])
        ls = []
        for u, i, j in batches:
            if use_fast:
                with oracle.fast():
                    ls.append(oracle.mf_train_step(oracle.LOSS_RUBIBCEBOTH, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs).copy())
            else:
                ls.append(oracle.mf_train_step(oracle.LOSS_RUBIBCEBOTH, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs).copy())
        outs.append((np.array(ls), Po, Qo, wo, wuo))
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-6)
    for a, b in zip(outs[1][1:], outs[0][1:]):
        np.testing.assert_allclose(a, b, atol=1e-6)
    # the second step of the checker on the same tables (persistent gradient scratch: the rows of step 1 are zero again)
    assert np.isfinite(outs[0][0]).all() and outs[0][0][1, 0] != outs[0][0][0, 0]
