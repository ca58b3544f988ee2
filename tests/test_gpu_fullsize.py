"""BASELINE.json-sized cases on the GPU (-m gpu): the oracle checks a slice it can finish in seconds,
the rest is covered by size-independent properties (split/shard invariance, idempotence, linearity)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from macr_amd import ops as _ops
    return _ops


@pytest.fixture(params=["f32", "bf16", "f16"])
def eval_filter(request, ops, monkeypatch):
    """run the test once per candidate filter of the listing pass (include/macr_hip.h MACR_EVAL_FILTER_*): the ranking
    must be the fp32 ranking bit for bit either way"""
    monkeypatch.setenv("MACR_EVAL_FILTER", request.param)       # what an Evaluator created by the test picks up
    ops.set_eval_filter(request.param)
    yield request.param
    ops.set_eval_filter("env")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


SYNTH128 = dict(n_users=40000, n_items=30011, d=128, batch=8192, n_train=1600000, n_test_users=5000, test_per_user=8,
                alpha=1e-3, beta=1e-3, lr=1e-3, regs=1e-5, c=40.0)


@pytest.mark.parametrize("workload", ["gowalla", "ml10m", "synthetic-d128"])
def test_fullsize_eval(ops, workload, eval_filter):
    """configs[1] (15 424 query users x 40 981 items), configs[2] shapes (13 878 x 8 790) and a d=128 catalogue as in
    configs[4]; c=40, K=20.  The oracle ranks every 97th user, properties cover all of them."""
    from macr_amd import synth
    from macr_amd.evaluator import Evaluator
    cfg = SYNTH128 if workload == "synthetic-d128" else synth.WORKLOADS[workload]
    d = cfg["d"]
    rs = np.random.RandomState(1)
    P = (rs.standard_normal((cfg["n_users"], d)) * 0.3).astype(np.float32)
    pop = np.sort(rs.standard_normal(cfg["n_items"]))[::-1].astype(np.float32)       # popular items have low ids
    Q = (rs.standard_normal((cfg["n_items"], d)) * 0.3).astype(np.float32)
    Q[:, 0] += pop
    P[:, 0] = np.abs(P[:, 0])                        # scores rise and fall with popularity: adversarial stream order
    w = (rs.standard_normal(d) * 0.3).astype(np.float32)
    wu = (rs.standard_normal(d) * 0.3).astype(np.float32)
    users, mask, gt = synth.eval_problem(cfg, seed=3)
    ev = Evaluator(mask, gt, cfg["n_items"], torch.device("cuda"))
    uid, Pd, Qd, wd, wud = dev(users), dev(P), dev(Q), dev(w), dev(wu)
    mcsr = oracle.csr_from_lists(mask)
    for kind in (0, 1):
        val, idx, cnt = ev.rank(kind, Pd, uid, Qd, 20, wd, wud, 40.0)
        sel = np.arange(0, len(users), 97)                                   # oracle on every 97th user
        sig_i = ops.branch_sigmoid(Qd, wd).cpu().numpy()
        sig_u = ops.branch_sigmoid(Pd, wud, uid).cpu().numpy()
        sub = oracle.csr_from_lists([mask[q] for q in sel])
        wv, wi, wc = oracle.score_topk(kind, P[users[sel]], Q, 20, sig_u[sel], sig_i, 40.0, sub)
        assert np.array_equal(idx.cpu().numpy()[sel], wi)
        assert np.array_equal(val.cpu().numpy()[sel].view(np.uint32), wv.view(np.uint32))
        # properties on ALL users: split-count invariance, sortedness, no masked item, idempotence
        sig_ud = dev(sig_u); sig_id = dev(sig_i)
        v1, i1 = ops.score_topk(kind, Pd, uid, Qd, 20, sig_ud, sig_id, 40.0, ev.mask, 0, n_splits=7)
        m1 = ops.topk_merge(v1, i1)
        assert torch.equal(m1[1], idx) and torch.equal(m1[0], val)
        v = val.cpu().numpy(); ix = idx.cpu().numpy()
        assert np.all(v[:, :-1] >= v[:, 1:])
        ties = v[:, :-1] == v[:, 1:]
        assert np.all(ix[:, :-1][ties] < ix[:, 1:][ties])
        for q in range(0, len(users), 501):
            assert not set(ix[q]) & set(mask[q])
        again = ops.topk_merge(val.unsqueeze(0).contiguous(), idx.unsqueeze(0).contiguous())
        assert torch.equal(again[1], idx)


@pytest.mark.parametrize("K", [50, 100])
def test_fullsize_eval_wide_K(ops, K):
    """--Ks beyond 32 at the Gowalla shape (15 424 query users x 40 981 items): the wide ranking against the oracle on every
    97th user, sortedness / mask / item-shard invariance on all of them"""
    from macr_amd import synth
    from macr_amd.evaluator import Evaluator
    cfg = synth.WORKLOADS["gowalla"]
    d = cfg["d"]
    rs = np.random.RandomState(2)
    P = (rs.standard_normal((cfg["n_users"], d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((cfg["n_items"], d)) * 0.3).astype(np.float32)
    Q[:, 0] += np.sort(rs.standard_normal(cfg["n_items"]))[::-1].astype(np.float32)
    P[:, 0] = np.abs(P[:, 0])
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    users, mask, gt = synth.eval_problem(cfg, seed=3)
    ev = Evaluator(mask, gt, cfg["n_items"], torch.device("cuda"))
    uid, Pd, Qd, wd, wud = dev(users), dev(P), dev(Q), dev(w), dev(wu)
    val, idx, cnt = ev.rank(1, Pd, uid, Qd, K, wd, wud, 40.0)
    sel = np.arange(0, len(users), 97)
    sig_i = ops.branch_sigmoid(Qd, wd).cpu().numpy()
    sig_u = ops.branch_sigmoid(Pd, wud, uid).cpu().numpy()
    sub = oracle.csr_from_lists([mask[q] for q in sel])
    wv, wi, wc = oracle.score_topk(1, P[users[sel]], Q, K, sig_u[sel], sig_i, 40.0, sub)
    assert np.array_equal(idx.cpu().numpy()[sel], wi)
    assert np.array_equal(val.cpu().numpy()[sel].view(np.uint32), wv.view(np.uint32))
    v = val.cpu().numpy(); ix = idx.cpu().numpy()
    assert np.all(v[:, :-1] >= v[:, 1:])
    ties = v[:, :-1] == v[:, 1:]
    assert np.all(ix[:, :-1][ties] < ix[:, 1:][ties])
    for q in range(0, len(users), 501):
        assert not set(ix[q]) & set(mask[q])
    # two item shards, merged
    half = cfg["n_items"] // 2
    sig_ud, sig_id = dev(sig_u), dev(sig_i)
    parts = []
    for a, b in ((0, half), (half, cfg["n_items"])):
        pv, pi = ops.score_topk(1, Pd, uid, Qd[a:b], K, sig_ud, sig_id[a:b].contiguous(), 40.0, ev.mask, a, 1)
        lv, li, _ = ops.topk_merge(pv, pi)
        parts.append((lv, li))
    mv, mi, _ = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, idx) and torch.equal(mv, val)
    ret = ev.test_mf(1, Pd, uid, Qd, [20, K], wd, wud, 40.0)
    assert 0.0 <= ret["recall"][0] <= ret["recall"][1] <= 1.0


@pytest.mark.parametrize("workload,kind", [("ml10m", 1), ("gowalla", 1), ("gowalla", 0)])
def test_fullsize_train_step(ops, workload, kind):
    """configs[1]/[2]: B=4096 / 8192 on the real table shapes, one step against the oracle."""
    from macr_amd import synth
    cfg = synth.WORKLOADS[workload]
    B, d = cfg["batch"], 64
    rs = np.random.RandomState(2)
    P = (rs.standard_normal((cfg["n_users"], d)) * 0.1).astype(np.float32)
    Q = (rs.standard_normal((cfg["n_items"], d)) * 0.1).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    u = rs.choice(cfg["n_users"], B, replace=False).astype(np.int32)
    i = (rs.zipf(1.3, B) - 1).clip(0, cfg["n_items"] - 1).astype(np.int32)
    j = rs.randint(0, cfg["n_items"], B).astype(np.int32)
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    state = ops.MFState(dev(P), dev(Q), dev(w), dev(wu),
                        ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B), B)
    want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
    got = state.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5)
    g_hip, g_orc = state.mQ.cpu().numpy() / 0.1, st.m[1] / 0.1
    np.testing.assert_allclose(g_hip, g_orc, rtol=5e-4, atol=2e-6 * np.abs(g_orc).max())
    # tables within 0.2 % of a step (the bound of the small-shape test, tests/test_gpu_ops.py), Adam slots like there
    for name, mine, theirs in (("P", state.P, Po), ("Q", state.Q, Qo), ("w", state.w, wo), ("wu", state.wu, wuo)):
        np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=0, atol=2e-3 * cfg["lr"], err_msg=name)
    for name, mine, theirs in (("mP", state.mP, st.m[0]), ("mQ", state.mQ, st.m[1]), ("vP", state.vP, st.v[0]),
                               ("vQ", state.vQ, st.v[1])):
        np.testing.assert_allclose(mine.cpu().numpy(), theirs, rtol=2e-4, atol=2e-6 * np.abs(theirs).max(), err_msg=name)


def test_yelp_size_propagation(ops):
    """configs[3]: N=69 716 nodes, nnz ~2.7 M, 2 layers, d=64.  Oracle on the full graph + linearity."""
    import scipy.sparse as sp
    from macr_amd import synth
    cfg = synth.WORKLOADS["yelp2018"]
    n_u, n_i = cfg["n_users"], cfg["n_items"]
    lists = synth.interaction_lists(n_u, n_i, cfg["n_train"] / n_u, seed=9)
    rows = np.repeat(np.arange(n_u), [len(l) for l in lists])
    cols = np.concatenate(lists)
    R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_u, n_i))
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5).astype(np.float32)
    dinv[np.isinf(dinv)] = 0
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A.sort_indices()
    rs = np.random.RandomState(0)
    E0 = rs.standard_normal((n_u + n_i, 64)).astype(np.float32)
    adj = ops.CSR.from_scipy(A, "cuda")
    got = ops.lgcn_propagate(adj, dev(E0), 2)
    want = oracle.lgcn_propagate(A.indptr, A.indices, A.data, E0, 2)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=3e-5, atol=3e-6)
    E1 = rs.standard_normal(E0.shape).astype(np.float32)
    lin = ops.lgcn_propagate(adj, dev(2.0 * E0 + E1), 2) - (2.0 * got + ops.lgcn_propagate(adj, dev(E1), 2))
    assert float(lin.abs().max()) < 2e-5
    # symmetric operator: <x, A y> == <A x, y>  (the backward pass reuses the forward kernel)
    x, y = dev(E0), dev(E1)
    lhs = float((x.double() * ops.lgcn_propagate(adj, y, 2).double()).sum())
    rhs = float((ops.lgcn_propagate(adj, x, 2).double() * y.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * abs(lhs) + 1e-3


def yelp_graph(cfg, seed=9):
    import scipy.sparse as sp
    from macr_amd import synth
    n_u, n_i = cfg["n_users"], cfg["n_items"]
    lists = synth.interaction_lists(n_u, n_i, cfg["n_train"] / n_u, seed=seed)
    rows = np.repeat(np.arange(n_u), [len(l) for l in lists])
    cols = np.concatenate(lists)
    R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_u, n_i))
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    deg = np.asarray(A.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5).astype(np.float32)
    dinv[np.isinf(dinv)] = 0
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocsr().astype(np.float32)
    A.sort_indices()
    return A


@pytest.mark.parametrize("kind", [1, 0])
def test_yelp_size_lightgcn_train_step(ops, kind):
    """configs[3], the WHOLE step: Yelp2018 shapes (N = 69 716 nodes, nnz ~2.7 M), 2 layers, d = 64, B = 4096:
    propagation, pair loss on propagated rows, backward through the propagation, ego-row regulariser, dense Adam --
    two consecutive steps against the oracle (LightGCN.py:288-309, :495-532 / :415-429, :525-528, :201 / :186)."""
    from macr_amd import synth
    cfg = synth.WORKLOADS["yelp2018"]
    n_u, n_i, d, B = cfg["n_users"], cfg["n_items"], 64, cfg["batch"]
    A = yelp_graph(cfg)
    rs = np.random.RandomState(4)
    T = (rs.standard_normal((n_u + n_i, d)) * 0.1).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    adj = ops.CSR.from_scipy(A, "cuda")
    state = ops.LGCNState(dev(T), n_u, n_i, dev(w), dev(wu), adj, 2,
                          ops.make_hyper(cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B), B)
    st = oracle.AdamState([T.shape, (d,), (d,)])
    To, wo, wuo = T.copy(), w.copy(), wu.copy()
    for t in range(2):
        u = rs.choice(n_u, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.3, B) - 1).clip(0, n_i - 1).astype(np.int32)          # popular positives: hot rows in the scatter
        j = rs.randint(0, n_i, B).astype(np.int32)
        want = oracle.lgcn_train_step(kind, n_u, n_i, 2, A.indptr, A.indices, A.data, u, i, j, To, wo, wuo, st,
                                      cfg["lr"], cfg["regs"], cfg["alpha"], cfg["beta"], B)
        got = state.step(kind, dev(u), dev(i), dev(j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, err_msg="step %d" % t)
        if t == 0:           # m = 0.1 * dense gradient of the ego table (every row: the propagation spreads it)
            g_hip, g_orc = state.mT.cpu().numpy() / 0.1, st.m[0] / 0.1
            np.testing.assert_allclose(g_hip, g_orc, rtol=5e-4, atol=2e-6 * np.abs(g_orc).max())
    np.testing.assert_allclose(state.mT.cpu().numpy(), st.m[0], rtol=5e-4, atol=2e-6 * np.abs(st.m[0]).max())
    np.testing.assert_allclose(state.vT.cpu().numpy(), st.v[0], rtol=1e-3, atol=2e-6 * np.abs(st.v[0]).max())
    np.testing.assert_allclose(state.T.cpu().numpy(), To, rtol=0, atol=4e-3 * cfg["lr"])
    if kind == 1:
        np.testing.assert_allclose(state.w.cpu().numpy(), wo, rtol=0, atol=4e-3 * cfg["lr"])
        np.testing.assert_allclose(state.wu.cpu().numpy(), wuo, rtol=0, atol=4e-3 * cfg["lr"])


def test_config4_shard_eval(ops, eval_filter):
    """configs[4] (10 M x 1 M, d = 128, item-sharded over 8 GPUs), ONE rank's share at full size: 20 000 query users
    against a 125 000-item shard whose global ids start at item_offset.  The oracle ranks every 97th user bit for
    bit; all users: the shard's result is independent of how the listing is split, sub-shards merge to the same
    lists (what the all-gather + merge does across ranks), no masked item is returned, ids carry the offset."""
    from macr_amd import synth
    from macr_amd.evaluator import Evaluator
    cfg = dict(n_users=200000, n_items=125000, d=128, n_train=200000 * 20, n_test_users=20000, test_per_user=10)
    d, off = 128, 3 * 125000                                   # rank 3 of 8
    rs = np.random.RandomState(5)
    P = (rs.standard_normal((cfg["n_users"], d)) * 0.2).astype(np.float32)
    Q = (rs.standard_normal((cfg["n_items"], d)) * 0.2).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    users, mask_local, _ = synth.eval_problem(cfg, seed=11)
    mask = [[off + x for x in row] for row in mask_local]      # masks hold GLOBAL ids
    Pd, Qd, wd, wud, uid = dev(P), dev(Q), dev(w), dev(wu), dev(users)
    sig_i = ops.branch_sigmoid(Qd, wd)
    sig_u = ops.branch_sigmoid(Pd, wud, uid)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    v, ix = ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Qd, 20, sig_u, sig_i, 40.0, mcsr, off)
    val, idx, cnt = ops.topk_merge(v, ix)
    sel = np.arange(0, len(users), 97)
    wv, wi, wc = oracle.score_topk(oracle.SCORE_RUBI_BOTH, P[users[sel]], Q, 20, sig_u.cpu().numpy()[sel],
                                   sig_i.cpu().numpy(), 40.0, oracle.csr_from_lists([mask[q] for q in sel]), item_offset=off)
    assert np.array_equal(idx.cpu().numpy()[sel], wi)
    assert np.array_equal(val.cpu().numpy()[sel].view(np.uint32), wv.view(np.uint32))
    ixn = idx.cpu().numpy()
    assert ixn.min() >= off and ixn.max() < off + cfg["n_items"]
    for q in range(0, len(users), 211):
        assert not set(ixn[q]) & set(mask[q])
    # two sub-shards of this shard, merged: the 1/2/4/8-rank invariance at shard scale
    half = cfg["n_items"] // 2 + 7
    parts = []
    for lo, hi in ((0, half), (half, cfg["n_items"])):
        pv, pi = ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Qd[lo:hi].contiguous(), 20, sig_u, sig_i[lo:hi].contiguous(),
                                40.0, mcsr, off + lo)
        mv, mi, _ = ops.topk_merge(pv, pi)
        parts.append((mv, mi))
    gv = torch.stack([p[0] for p in parts]); gi = torch.stack([p[1] for p in parts])
    m2 = ops.topk_merge(gv, gi)
    assert torch.equal(m2[1], idx) and torch.equal(m2[0], val)


def test_config4_shard_size_training_step(ops):
    """configs[4] training at ONE rank's size: 1 250 000 user rows + 125 000 item rows, d = 128 (what a rank of 8 holds
    of the 10 M x 1 M tables), B = 8192, through the row-shard entry points (macr_shard_*: gather, forward, (B,B) row
    blocks, backward into the staging buffer, radix sort of the references, segment reduce, dense Adam over the
    shard) -- two steps against the oracle: losses 1e-5, the gradient (first-step m) and the updated rows the batch
    touched, plus untouched rows (dense Adam moves every row: bitwise equal to a zero-gradient step of the oracle)."""
    from macr_amd import sharded_train
    n_users, n_items, d, B = 1_250_000, 125_000, 128, 8192
    rs = np.random.RandomState(8)
    P = (rs.standard_normal((n_users, d)) * 0.05).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.05).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.2).astype(np.float32), (rs.standard_normal(d) * 0.2).astype(np.float32)
    lr, decay, alpha, beta = 1e-3, 1e-5, 1e-3, 1e-3
    hyper = ops.make_hyper(lr, decay, alpha, beta, B)
    kind = ops.LOSS_RUBIBCEBOTH
    model = sharded_train.RowShardedMF(dev(P), dev(Q), dev(w), dev(wu),
                                       sharded_train.HipBackend(kind, d, hyper, torch.device("cuda")), rank=0, world=1)
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P, Q, w.copy(), wu.copy()                    # the oracle updates in place
    touched_u, touched_i = set(), set()
    for t in range(2):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.2, B) % n_items).astype(np.int32)            # Zipf positives: hot rows in the segment reduce
        j = rs.randint(0, n_items, B).astype(np.int32)
        touched_u.update(u.tolist()); touched_i.update(i.tolist()); touched_i.update(j.tolist())
        want = oracle.mf_train_step(oracle.LOSS_RUBIBCEBOTH, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, B)
        got = model.step(dev(u), dev(i), dev(j)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, err_msg="step %d" % t)
    ru = np.fromiter(touched_u, np.int64)[:4000]
    ri = np.fromiter(touched_i, np.int64)[:4000]
    tol = 2e-3 * lr * 2
    np.testing.assert_allclose(model.P[torch.from_numpy(ru).cuda()].cpu().numpy(), Po[ru], rtol=0, atol=tol)
    np.testing.assert_allclose(model.Q[torch.from_numpy(ri).cuda()].cpu().numpy(), Qo[ri], rtol=0, atol=tol)
    np.testing.assert_allclose(model.mQ[torch.from_numpy(ri).cuda()].cpu().numpy(), st.m[1][ri], rtol=1e-3,
                               atol=2e-6 * np.abs(st.m[1][ri]).max())
    # rows no batch touched: m = v = 0, theta unchanged by a zero-gradient Adam step -- on both sides, bit for bit
    free = np.setdiff1d(np.arange(0, n_users, 997), np.fromiter(touched_u, np.int64))
    assert np.array_equal(model.P[torch.from_numpy(free).cuda()].cpu().numpy(), Po[free])
    assert float(model.mP[torch.from_numpy(free).cuda()].abs().max()) == 0.0
    np.testing.assert_allclose(model.w.cpu().numpy(), wo, rtol=0, atol=tol)
    assert int(model.tP.sum()) == 0 and int(model.tQ.sum()) == 0


@pytest.mark.parametrize("workload", ["gowalla", "ml10m"])
def test_fullsize_seeded_rankings_at_every_staleness(ops, workload, eval_filter):
    """Threshold seeds at full size (512 resident workgroups, 61 blocks of 256 queries on Gowalla shapes): the ranking
    leaves its best candidates, the tables then move by a little, by a lot, or are replaced, and the seeded ranking
    (good seeds; seeds that are stale for SOME query blocks: early stop of single blocks, repair round for those; seeds
    that are stale everywhere: every block stops within a few tiles) must equal the unseeded ranking of the same
    tables in every id and every score bit, for all users.  stats tell which path ran."""
    from macr_amd import synth
    cfg = synth.WORKLOADS[workload]
    d, K = cfg["d"], 20
    rs = np.random.RandomState(4)
    P = (rs.standard_normal((cfg["n_users"], d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((cfg["n_items"], d)) * 0.3).astype(np.float32)
    Q[:, 0] += np.sort(rs.standard_normal(cfg["n_items"]))[::-1].astype(np.float32)
    P[:, 0] = np.abs(P[:, 0])
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    users, mask, gt = synth.eval_problem(cfg, seed=3)
    U = len(users)
    mcsr = ops.CSR.from_lists(mask, "cuda")
    uid, Pd, wd, wud = dev(users), dev(P), dev(w), dev(wu)
    sig_u = ops.branch_sigmoid(Pd, wud, uid)
    seeds = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device="cuda")
    stats = torch.zeros(2, dtype=torch.int32, device="cuda")
    Qd = dev(Q)
    ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Qd, K, sig_u, ops.branch_sigmoid(Qd, wd), 40.0, mcsr, seed_out=seeds)
    blocks = (U + 255) // 256
    seen = []
    for name, noise, rows in (("drift", 0.003, None), ("partly stale", 0.6, slice(0, cfg["n_items"] // 9)), ("replaced", None, None)):
        Q2 = Q.copy()
        if noise is None:
            Q2 = (rs.standard_normal(Q.shape) * 0.3).astype(np.float32)
        elif rows is None:
            Q2 += rs.standard_normal(Q.shape).astype(np.float32) * noise
        else:
            Q2[rows] += rs.standard_normal(Q2[rows].shape).astype(np.float32) * noise
        Q2d = dev(Q2)
        sig_i = ops.branch_sigmoid(Q2d, wd)
        v0, i0 = ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Q2d, K, sig_u, sig_i, 40.0, mcsr)
        sd = seeds.clone()
        v1, i1 = ops.score_topk(ops.SCORE_RUBI_BOTH, Pd, uid, Q2d, K, sig_u, sig_i, 40.0, mcsr, seed=sd, seed_out=sd, stats=stats)
        assert torch.equal(i0, i1), name
        assert torch.equal(v0.view(torch.int32), v1.view(torch.int32)), name
        assert torch.equal(sd[:, :K], i1[0]), name
        seen.append(stats.tolist())
    assert seen[0] == [0, 0], seen                        # seeds that hold: no block listed twice
    assert seen[2][0] == blocks and seen[2][1] == 0, seen  # replaced tables: every block re-listed, no exact fallback
    assert seen[1][1] == 0, seen


def test_a_table_beyond_four_gib(ops):
    """configs[4] on one GPU: a user table of 9 M x 128 fp32 = 4.6 GB, every batch user beyond the 2^32-byte mark of the
    table (and of its Adam slots and gradient scratch).  The oracle runs on the COMPACT problem -- the rows the batch
    refers to, renumbered -- which is the same step: TF's dense Adam leaves a row without gradient and with zero slots
    exactly where it was.  Touched rows against the oracle, a sample of untouched rows (both sides of the 4 GiB mark)
    bit-unchanged; once through macr_mf_train_step, once through the row-sharded entry points (world = 1)."""
    from macr_amd import sharded_train
    n_users, n_items, d, B = 9_000_000, 50_000, 128, 4096
    assert n_users * d * 4 > 2 ** 32
    lr, decay, alpha, beta = 1e-3, 1e-5, 1e-3, 1e-3
    gen = torch.Generator(device="cuda").manual_seed(11)
    rs = np.random.RandomState(12)
    first_high = 2 ** 32 // (d * 4) + 1                                    # first row that lies entirely beyond 4 GiB
    u = np.sort(rs.choice(np.arange(first_high, n_users), B, replace=False)).astype(np.int32)
    rs.shuffle(u)
    i = (rs.zipf(1.3, B) % n_items).astype(np.int32)
    j = rs.randint(0, n_items, B).astype(np.int32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    hyper = ops.make_hyper(lr, decay, alpha, beta, B)
    probe = np.concatenate([np.arange(0, 64), np.arange(first_high - 32, first_high + 32), rs.randint(0, n_users, 256)])
    probe = np.setdiff1d(probe, u).astype(np.int64)
    for path in ("single", "sharded"):
        P = (torch.randn((n_users, d), generator=gen, device="cuda", dtype=torch.float32) * 0.05).contiguous()
        Q = (torch.randn((n_items, d), generator=gen, device="cuda", dtype=torch.float32) * 0.05).contiguous()
        # compact oracle problem: the batch's user rows 0..B-1
        Pc = np.ascontiguousarray(P[torch.from_numpy(u.astype(np.int64)).cuda()].cpu().numpy(), dtype=np.float32)
        Qc = np.ascontiguousarray(Q.cpu().numpy(), dtype=np.float32).copy()
        before = P[torch.from_numpy(probe).cuda()].clone()
        st = oracle.AdamState([Pc.shape, Qc.shape, (d,), (d,)])
        wo, wuo = w.copy(), wu.copy()
        want = oracle.mf_train_step(1, np.arange(B, dtype=np.int32), i, j, Pc, Qc, wo, wuo, st, lr, decay, alpha, beta, B)
        if path == "single":
            state = ops.MFState(P, Q, dev(w), dev(wu), hyper, B)
            got = state.step(1, dev(u), dev(i), dev(j)).cpu().numpy()
            Pn, Qn = state.P, state.Q
        else:
            model = sharded_train.RowShardedMF(None, None, dev(w), dev(wu), sharded_train.HipBackend(1, d, hyper, torch.device("cuda")),
                                               rank=0, world=1, shards=(P, Q, n_users, n_items))
            got = model.step(dev(u), dev(i), dev(j)).cpu().numpy()
            Pn, Qn = model.P, model.Q
        np.testing.assert_allclose(got, want, rtol=1e-5, err_msg=path)
        np.testing.assert_allclose(Pn[torch.from_numpy(u.astype(np.int64)).cuda()].cpu().numpy(), Pc, rtol=0, atol=2e-3 * lr, err_msg=path)
        np.testing.assert_allclose(Qn.cpu().numpy(), Qc, rtol=0, atol=2e-3 * lr, err_msg=path)
        assert torch.equal(Pn[torch.from_numpy(probe).cuda()], before), path       # no gradient, zero slots: untouched
        del P, Q, Pn, Qn
        if path == "single":
            del state
        else:
            del model
        torch.cuda.empty_cache()
