"""The lazy dense Adam pass (include/macr_hip.h: macr_lazy_adam; tf.train.AdamOptimizer's every-row-every-step update of
macr_mf/model.py:74,:95 blocked in time) against the per-step dense pass of the same library: the SAME bits -- losses of every
step, tables and slots after a flush -- for every period, table shape and reference-list path, whenever the tables are read."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(n_users, n_items, d, periods, kind, seed=5):
    from macr_amd import ops, sharded_train
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(seed)
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = []
    for k in periods:
        hyper = ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024)
        out.append(sharded_train.RowShardedMF(t(P), t(Q), t(w), t(wu), sharded_train.HipBackend(kind, d, hyper, dev), rank=0, world=1,
                                              lazy_period=k))
    return out, rs


def _batch(rs, n_users, n_items, B, hot=(40, 20)):
    """every path of the gradient sums that is deterministic: most rows once or a few times (the indexed staging rows), item 1
    forty and item 2 twenty times (one work item of k_seg_sum each, through gQ).  Rows with more than 64 references are summed
    by several work items whose atomic adds arrive in any order -- two runs of the DENSE pass differ in their last bit there,
    so a bit-for-bit comparison cannot include them (tools/lazy_diag.py shows exactly that row and nothing else)."""
    dev = torch.device("cuda", 0)
    u = rs.choice(n_users, B, replace=B > n_users).astype(np.int32)
    i = rs.randint(0, n_items, B).astype(np.int32)
    i[:hot[0]], i[hot[0]:hot[0] + hot[1]] = 1, 2
    j = rs.randint(3, n_items, B).astype(np.int32)
    return [torch.from_numpy(a).to(dev) for a in (u, i, j)]


def _lockstep(dense, others, u, i, j):
    """One step of `dense` and of every model in `others` on the SAME gradients: the per-pair forward / (B,B) / backward run once
    (on dense's rows) and their results -- staged gradient rows, branch-vector partials, lr_t: the step's workspace -- are
    copied to the others before each applies its own optimizer pass.  Two free-running models do not stay bit-identical
    whatever their optimizer: the branch-vector gradient partials are summed by atomic adds in arrival order
    (k_pair_bwd_stage), so w and everything after it differ in the last bit from run to run (tools/lazy_diag.py).
    What a model gathers for the step must already equal what the dense model gathers."""
    be = dense.backend
    rows3 = be.gather(dense, u, i, j)
    for m in others:
        got = m.backend.gather(m, u, i, j)
        assert torch.equal(got, rows3), "gathered rows differ: max |diff| %g" % float((got - rows3).abs().max())
    be.forward_and_bxb(dense, rows3, 0, 1)
    losses, _ = be.backward(dense, rows3)
    for m in others:
        m.backend.ws.copy_(be.ws)
    be.apply(dense, u, i, j)
    for m in others:
        m.backend.apply(m, u, i, j)
        m._stale = m.lazy_period > 1
    return losses


def _same_state(a, b):
    for name in ("P", "Q", "mP", "vP", "mQ", "vQ", "w", "wu", "mw", "vw", "mwu", "vwu"):
        x, y = getattr(a, name), getattr(b, name)
        assert torch.equal(x, y), "%s differs: max |diff| %g" % (name, float((x - y).abs().max()))
    assert int(a.tP.abs().sum()) == 0 and int(a.tQ.abs().sum()) == 0          # every flag consumed
    assert float(a.gP.abs().max()) == 0.0 and float(a.gQ.abs().max()) == 0.0   # every gradient row consumed and cleared


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_lazy_pass_equals_the_dense_pass_bit_for_bit(d):
    """periods 2, 5 and 64 (more than the steps: nothing but the batch's rows and a 64th of the chunks moves before the flush)
    against the dense pass on the same gradients: the rows every step gathers and, after a flush, every table and slot"""
    from macr_amd import ops
    n_users, n_items, B, steps = 5003, 1201, 1500, 23
    periods = [1, 2, 5, 64]
    models, rs = _models(n_users, n_items, d, periods, ops.LOSS_RUBIBCEBOTH)
    assert [m.lazy_period for m in models] == periods
    for step in range(steps):
        _lockstep(models[0], models[1:], *_batch(rs, n_users, n_items, B))
        if step == 11:                                   # a reader in the middle of training: flush, compare, carry on
            for m in models[1:]:
                assert m._stale
                _same_state(m, models[0])
                assert not m._stale
    for m in models[1:]:
        _same_state(m, models[0])
        assert int(m.stP.min()) == steps and int(m.stQ.max()) == steps        # every row at the last step


def test_lazy_training_free_running():
    """the whole step through RowShardedMF.step (every model its own forward and backward): losses and tables agree with the
    dense pass to the last bits the atomically summed branch-vector gradients leave"""
    from macr_amd import ops
    n_users, n_items, d, B = 5003, 1201, 64, 1500
    (dense, lazy), rs = _models(n_users, n_items, d, [1, 4], ops.LOSS_RUBIBCEBOTH)
    for step in range(12):
        u, i, j = _batch(rs, n_users, n_items, B)
        a, b = dense.step(u, i, j), lazy.step(u, i, j)
        assert float((a / b - 1).abs().max()) < 1e-6, (step, a, b)
    for name in ("P", "Q", "w", "wu"):
        assert float((getattr(dense, name) - getattr(lazy, name)).abs().max()) < 1e-6, name


def test_lazy_pass_leaves_untouched_rows_alone_between_sweeps():
    """what the pass is for: with a long period a step visits the batch's rows and its K-th of the chunks -- every other row
    keeps its bits and its stamp until its sweep or the flush"""
    from macr_amd import ops
    n_users, n_items, d, B, K = 20000, 3000, 64, 512, 64
    (m,), rs = _models(n_users, n_items, d, [K], ops.LOSS_RUBIBCEBOTH)
    P0 = m._P.clone()
    visited = torch.zeros(n_users, dtype=torch.bool, device="cuda")
    rows_per_chunk = 1024 // (d // 4)                           # kAdamVecPerBlock float4 per chunk
    chunk_of = torch.arange(n_users, device="cuda") // rows_per_chunk
    for step in range(1, 4):
        u, i, j = _batch(rs, n_users, n_items, B)
        m.step(u, i, j)
        visited[u.long()] = True
        visited |= (chunk_of % K) == (step % K)
    assert torch.equal(m.stP != 0, visited)
    assert 3 * 512 * 0.9 < int(visited.sum()) < 3 * (512 + 6 * rows_per_chunk)      # of 20 000 rows
    assert torch.equal(m._P[~visited], P0[~visited])
    m.flush()
    assert int((m.stP != 3).sum()) == 0


@pytest.mark.parametrize("kind_name", ["LOSS_NORMALBCE", "LOSS_RUBIBCE"])
def test_lazy_pass_other_losses_and_the_unfused_reference_path(kind_name, monkeypatch):
    """normalbce (no branch vectors) and rubibce; the second half of the run with MACR_SEG_UNFUSED=1 (every gradient row
    through gP / gQ instead of the indexed staging rows; there a row's references are summed per 16-entry chunk of the sorted list
    and the chunks' sums meet by atomic adds -- in any order, so a bit-for-bit run keeps every row within two chunks)"""
    from macr_amd import ops
    kind = getattr(ops, kind_name)
    n_users, n_items, d, B = 3001, 900, 64, 777
    (dense, lazy), rs = _models(n_users, n_items, d, [1, 3], kind)
    for step in range(14):
        if step == 7:
            monkeypatch.setenv("MACR_SEG_UNFUSED", "1")
        _lockstep(dense, [lazy], *_batch(rs, n_users, n_items, B, hot=(40, 20) if step < 7 else (16, 12)))
    _same_state(lazy, dense)


def test_lazy_rows_reads_a_row_as_of_now_and_writes_nothing():
    from macr_amd import ops
    n_users, n_items, d, B = 4000, 1000, 128, 300
    (dense, lazy), rs = _models(n_users, n_items, d, [1, 7], ops.LOSS_RUBIBCEBOTH)
    for step in range(5):
        _lockstep(dense, [lazy], *_batch(rs, n_users, n_items, B))
    before = [t.clone() for t in (lazy._P, lazy._mP, lazy._vP, lazy.stP)]
    rows = torch.from_numpy(np.random.RandomState(1).randint(-1, n_users, 2000).astype(np.int32)).cuda()
    got = lazy.backend.lazy_rows(lazy, "P", rows)
    want = torch.where((rows >= 0).unsqueeze(1), dense.P[rows.clamp(min=0).long()], torch.zeros((), device="cuda"))
    assert torch.equal(got, want)
    for a, b in zip(before, (lazy._P, lazy._mP, lazy._vP, lazy.stP)):
        assert torch.equal(a, b)
    assert int((lazy.stP != 5).sum()) > 1000                    # (most rows really were behind)
    # what an evaluation touches: the item table (brought up to date: the whole catalogue is scored) and the query users' rows
    # (read as of now) -- the user table stays behind
    assert torch.equal(lazy.Q, dense.Q) and int((lazy.stQ != 5).sum()) == 0
    idx = rows[rows >= 0][:500]
    assert torch.equal(lazy.rows("P", idx), dense.P[idx.long()])
    assert int((lazy.stP != 5).sum()) > 1000 and lazy._stale_tabs == {"P"}
    assert torch.equal(lazy.P, dense.P) and not lazy._stale


def test_lazy_period_rule_and_argument_checks():
    from macr_amd import _lib, ops, sharded_train
    assert sharded_train.lazy_period_for(29858 + 40981, 64) == 1            # Gowalla: dense every step
    assert sharded_train.lazy_period_for(11_000_000, 128) == 64             # BASELINE configs[4] on one GPU
    assert sharded_train.lazy_period_for(1_375_000, 128) == 16              # ... on eight
    L = _lib.lib()
    hyper = ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024)
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    import ctypes
    p = ops._ptr(buf)
    for period in (0, 65):
        lz = _lib.LazyAdam(p, p, p, period)
        rc = L.macr_lazy_flush(64, 4, 4, p, p, p, p, p, p, ctypes.byref(hyper), ctypes.byref(lz), None)
        assert rc == _lib.E_INVALID and b"period" in L.macr_last_error()


# ---------------------------------------------------------------------------------------------------------------------
# The MF step in deferred mode (macr_mf_train_step_lazy): the lazy pass rides in the (B,B) launch, the forward kernel looks
# stamp .. pending step ahead in registers and marks its rows for the pass.
# ---------------------------------------------------------------------------------------------------------------------
def _mf_states(n_users, n_items, d, B, periods, seed=9):
    from macr_amd import ops
    rs = np.random.RandomState(seed)
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()
    return [ops.MFState(t(P), t(Q), t(w), t(wu), ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024), B, lazy_period=k) for k in periods], rs


_MF_NAMES = ("P", "Q", "w", "wu", "mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu")


@pytest.mark.parametrize("d,kind_name", [(32, "LOSS_RUBIBCEBOTH"), (64, "LOSS_RUBIBCEBOTH"), (128, "LOSS_RUBIBCE"), (256, "LOSS_RUBIBCEBOTH")])
def test_mf_lazy_sequence_equals_the_dense_sequence_bit_for_bit(d, kind_name):
    """Batches without a repeated row and with at most eight backward blocks: every atomic add of the step lands on a zero (one
    per gradient element, one per branch-vector partial slot), so a deferred sequence is reproducible bit for bit -- and the
    lazy form (periods 2, 3, 7) must reproduce the dense form: every step's losses, and the tables whenever they are flushed."""
    from macr_amd import ops
    kind = getattr(ops, kind_name)
    n_users, n_items, B, steps = 2000, 1500, 128, 26
    states, rs = _mf_states(n_users, n_items, d, B, [1, 2, 3, 7])
    for step in range(steps):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        ij = rs.choice(n_items, 2 * B, replace=False).astype(np.int32)
        b = [torch.from_numpy(a).cuda() for a in (u, ij[:B], ij[B:])]
        losses = [s.step(kind, *b, defer=True).clone() for s in states]
        assert states[1]._seq_lazy is not None and states[0]._seq_lazy is None
        for l in losses[1:]:
            assert torch.equal(l, losses[0]), (step, l, losses[0])
        if step in (9, steps - 1):
            for s in states:
                s.flush()
            for s in states[1:]:
                for name in _MF_NAMES:
                    assert torch.equal(getattr(s, name), getattr(states[0], name)), (step, name)
                assert int(s.tP.abs().sum()) == 0 and int(s.tQ.abs().sum()) == 0 and float(s.gP.abs().max()) == 0.0
                st, sp, sq = s._lazy_bufs
                assert int(sp.min()) == step + 1 and int(sq.max()) == step + 1


@pytest.mark.parametrize("B,d,n_users,n_items,K", [(1024, 64, 13485, 744, 4), (4096, 64, 29858, 40981, 4), (257, 32, 5000, 3000, 2),
                                                   (9000, 64, 120000, 30000, 8), (4096, 128, 50000, 9000, 16)])
def test_mf_lazy_sequence_on_real_batches(B, d, n_users, n_items, K):
    """popularity-skewed batches (duplicates, the bucketed backward, B > 8192: the staged backward): the lazy sequence against the
    dense sequence and against complete steps, within what the order of the float atomics leaves (the bounds of
    test_mf_deferred_random_shapes)"""
    from macr_amd import ops
    kind = ops.LOSS_RUBIBCEBOTH
    (dense, lazy, eager), rs = _mf_states(n_users, n_items, d, B, [1, K, 1])
    for step in range(11):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.3, B) % n_items).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        b = [torch.from_numpy(a).cuda() for a in (u, i, j)]
        a_, l_, e_ = dense.step(kind, *b, defer=True), lazy.step(kind, *b, defer=True), eager.step(kind, *b)
        assert lazy._seq_lazy is not None
        np.testing.assert_allclose(l_.cpu().numpy(), a_.cpu().numpy(), rtol=5e-6)
        np.testing.assert_allclose(l_.cpu().numpy(), e_.cpu().numpy(), rtol=5e-6)
        if step == 5:
            lazy.flush()                                 # a reader in mid-sequence; the next step starts a new sequence
    dense.flush(); lazy.flush()
    for name in _MF_NAMES:
        x, y, z = (getattr(s, name).cpu().numpy() for s in (lazy, dense, eager))
        np.testing.assert_allclose(x, y, rtol=3e-4, atol=1e-7 + 2e-5 * np.abs(y).max(), err_msg=name)
        np.testing.assert_allclose(x, z, rtol=3e-4, atol=1e-7 + 2e-5 * np.abs(z).max(), err_msg=name)
    assert float(lazy.gP.abs().max()) == 0.0 and float(lazy.gQ.abs().max()) == 0.0
    assert int(lazy.tP.sum()) == 0 and int(lazy.tQ.sum()) == 0


def test_mf_lazy_step_that_completes_in_its_call():
    """a lazy sequence ended by a step without MACR_STEP_DEFER: the step's own pass and the catch-up of every row run in the call"""
    from macr_amd import ops
    kind = ops.LOSS_RUBIBCEBOTH
    n_users, n_items, d, B = 2000, 1500, 64, 128
    (dense, lazy), rs = _mf_states(n_users, n_items, d, B, [1, 3])
    for step in range(8):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        ij = rs.choice(n_items, 2 * B, replace=False).astype(np.int32)
        b = [torch.from_numpy(a).cuda() for a in (u, ij[:B], ij[B:])]
        defer = step < 7
        assert torch.equal(dense.step(kind, *b, defer=defer), lazy.step(kind, *b, defer=defer))
    assert lazy.pending_B == 0 and dense.pending_B == 0
    for name in _MF_NAMES:
        assert torch.equal(getattr(lazy, name), getattr(dense, name)), name


def test_mf_lazy_period_rule():
    from macr_amd import ops
    assert ops.mf_lazy_period_for(29858 + 40981, 64, 4096) == 1         # Gowalla: measured slower lazily (profiles/r05_lazy_mf_ab.txt)
    assert ops.mf_lazy_period_for(69878 + 10677, 64, 8192) == 1         # ML-10M
    assert ops.mf_lazy_period_for(300 + 50, 64, 96) == 1
    assert ops.mf_lazy_period_for(11_000_000, 128, 8192) == 45          # tables whose dense pass is 5.5 ms beside a 61 us (B,B) term
