"""Row-sharded training, host logic on CPU (gloo, world size 2): macr_amd.sharded_train.RowShardedMF with the ORACLE
as the device half (it is the checker, allowed in tests).  Ownership of rows, the all-reduce gather, the local apply
and the branch-vector broadcast must reproduce the single-process oracle step exactly (the oracle is deterministic)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from macr_amd import sharded_train


def test_row_ranges_partition_the_tables():
    for n, w in ((13485, 2), (10_000_000, 8), (7, 8), (1_000_000, 8)):
        prev = 0
        for r in range(w):
            lo, hi = sharded_train.row_range(n, r, w)
            assert lo == prev and hi >= lo
            prev = hi
        assert prev == n


def test_both_layouts_give_every_row_exactly_one_owner():
    for layout in ("interleaved", "range"):
        for n, w in ((13485, 2), (101, 3), (7, 8), (64, 8), (1, 3)):
            seen = np.zeros(n, np.int64)
            sizes = []
            for r in range(w):
                own = sharded_train.Owned(n, r, w, layout)
                ids = own.global_ids().numpy()
                assert len(ids) == own.n and (ids < n).all()
                seen[ids] += 1
                mask, loc = own.local_of(torch.arange(n))
                assert np.array_equal(np.nonzero(mask.numpy())[0], ids) and np.array_equal(loc[mask].numpy(), np.arange(own.n))
                assert torch.equal(own.take(torch.arange(n)), torch.from_numpy(ids))
                sizes.append(own.n)
            assert (seen == 1).all() and max(sizes) - min(sizes) <= 1
    # interleaved: the hot low ids do not share a rank
    assert [sharded_train.Owned(1000, r, 4).global_ids()[0].item() for r in range(4)] == [0, 1, 2, 3]


def _vp(a):
    import ctypes
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleBackend(object):
    """The device half played by oracle/macr_oracle.c on numpy views of the shard tensors."""

    def __init__(self, kind, d, lr, decay, alpha, beta, bs):
        import oracle
        self.o, self.kind, self.d = oracle, kind, d
        self.lr, self.decay, self.alpha, self.beta, self.bs = lr, decay, alpha, beta, bs
        self.power = np.asarray([0.9, 0.999], np.float32)

    def gather(self, shard, u, i, j):
        B = len(u)
        rows3 = np.zeros((3, B, self.d), np.float32)
        P, Q = shard.P.numpy(), shard.Q.numpy()
        for role, idx, tab, owner in ((0, u, P, shard.own_u), (1, i, Q, shard.own_i), (2, j, Q, shard.own_i)):
            own, loc = owner.local_of(idx.long())
            own, loc = own.numpy(), loc.numpy()
            rows3[role, own] = tab[loc[own]]
        return torch.from_numpy(rows3)

    def forward_and_bxb(self, shard, rows3, rank, world):
        return None                                   # the oracle evaluates the whole (B,B) term in backward()

    def backward(self, shard, rows3):
        o = self.o
        eu, ei, ej = (np.ascontiguousarray(rows3[k].numpy()) for k in range(3))
        res = o.pair_loss_grad(self.kind, eu, ei, ej, shard.w.numpy(), shard.wu.numpy(), self.alpha, self.beta)
        B, d = eu.shape
        deu, dei, dej = res["deu"], res["dei"], res["dej"]
        reg = o.lib().orc_l2_reg(B, d, eu, ei, ej, self.decay, self.bs, _vp(deu), _vp(dei), _vp(dej))      # adds coef*row to the gradients
        self.stage = (deu, dei, dej)
        self.gw = np.stack([res["dw"], res["dwu"]])
        losses = torch.tensor([res["mf"] + reg, res["mf"], reg], dtype=torch.float32)
        if self.kind == o.LOSS_NORMALBCE:
            return losses, None                       # no branch vectors: nothing to broadcast
        return losses, torch.from_numpy(self.gw)

    def apply(self, shard, u, i, j):
        o = self.o
        lr_t = o.lib().orc_adam_lr_t(self.lr, self.power)
        gP, gQ = np.zeros_like(shard.P.numpy()), np.zeros_like(shard.Q.numpy())
        for grad, idx, g, owner in ((self.stage[0], u, gP, shard.own_u), (self.stage[1], i, gQ, shard.own_i),
                                    (self.stage[2], j, gQ, shard.own_i)):
            own, loc = owner.local_of(idx.long())
            own, loc = own.numpy(), loc.numpy()
            for t in range(len(loc)):                 # batch order, like orc_scatter_add_rows
                if own[t]:
                    g[loc[t]] += grad[t]
        adam = lambda th, m, v, gr: o.lib().orc_adam_dense(th.numpy(), m.numpy(), v.numpy(), _vp(np.ascontiguousarray(gr)),
                                                           th.numel(), lr_t, 0.9, 0.999, 1e-8)
        adam(shard.P, shard.mP, shard.vP, gP)
        adam(shard.Q, shard.mQ, shard.vQ, gQ)
        if self.kind != o.LOSS_NORMALBCE:
            adam(shard.w, shard.mw, shard.vw, self.gw[0])
        if self.kind == o.LOSS_RUBIBCEBOTH:
            adam(shard.wu, shard.mwu, shard.vwu, self.gw[1])
        self.power *= np.asarray([0.9, 0.999], np.float32)


def _problem():
    rs = np.random.RandomState(12)
    n_users, n_items, d, B = 301, 77, 16, 96
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    batches = []
    for _ in range(3):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.3, B) % n_items).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        batches.append((u, i, j))
    return P, Q, w, wu, batches


HYP = dict(lr=1e-3, decay=1e-5, alpha=1e-2, beta=1e-3, bs=64)


def _worker(rank, world, port, q, kind_name="rubibceboth", layout="interleaved"):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, Q, w, wu, batches = _problem()
        kind = {"rubibceboth": oracle.LOSS_RUBIBCEBOTH, "rubibce": oracle.LOSS_RUBIBCE, "normalbce": oracle.LOSS_NORMALBCE}[kind_name]
        model = sharded_train.RowShardedMF(torch.from_numpy(P), torch.from_numpy(Q), torch.from_numpy(w), torch.from_numpy(wu),
                                           OracleBackend(kind, P.shape[1], **HYP), layout=layout)
        Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
        st = oracle.AdamState([P.shape, Q.shape, w.shape, wu.shape])
        ok = True
        for u, i, j in batches:
            got = model.step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j)).numpy()
            want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, HYP["lr"], HYP["decay"], HYP["alpha"], HYP["beta"],
                                        HYP["bs"])
            ok = ok and np.allclose(got, want, rtol=1e-6)
        Pf, Qf = model.full_tables()
        ok = ok and np.array_equal(Pf.numpy(), Po) and np.array_equal(Qf.numpy(), Qo)
        ok = ok and np.array_equal(model.w.numpy(), wo) and np.array_equal(model.wu.numpy(), wuo)
        ok = ok and model.P.shape[0] == sharded_train.Owned(P.shape[0], rank, world, layout).n
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_world(world, kind_name, layout):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind_name, layout)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)]


def test_row_sharded_training_gloo_world2():
    _run_world(2, "rubibceboth", "interleaved")


def test_row_sharded_training_gloo_world3_uneven_shards_normalbce():
    """three ranks (301 and 77 rows: shards of unequal size), the loss without a (B,B) term: one collective per step"""
    _run_world(3, "normalbce", "interleaved")


def test_row_sharded_training_gloo_world3_ranges_rubibce():
    _run_world(3, "rubibce", "range")


# ---------------------------------------------------------------------------------------------------------------------
# The SPLIT step (RowShardedMF.step_split): rows travel to the rank whose slice of the batch needs them and gradient rows
# travel back to their owners, by two all-to-alls.  The rig below plays the device half with the oracle: the routing, both
# exchanges, the placement of the received rows and the owners' apply must reproduce the single-process oracle step exactly.
# (The slice kernels themselves are checked on the GPU: tests/test_gpu_product.py::test_row_sharded_split_step_two_ranks_one_gpu.)
class SplitOracleBackend(OracleBackend):
    def slice_of(self, B, rank, world):
        return sharded_train.row_range(B, rank, world)

    def forward_slice(self, shard, B, t0, rows3_slice):
        n = rows3_slice.shape[1]
        self.region = torch.zeros((3, B, self.d), dtype=torch.float32)
        self.region[:, t0:t0 + n] = rows3_slice                   # "forward state of the slice": here simply its rows
        return self.region                                        # summed over the ranks by the caller (x + 0 = x)

    def bxb(self, B, rank, world):
        return torch.zeros(1)

    def backward_slice(self, shard, B, t0, rows3_slice):
        n = rows3_slice.shape[1]
        losses, branch = self.backward(shard, self.region)        # the oracle's gradients of the whole batch
        full = [torch.from_numpy(g) for g in self.stage]
        stage_slice = torch.stack([g[t0:t0 + n] for g in full]).contiguous()
        if shard.rank != 0:
            self.gw[:] = 0                                        # the branch-vector partials are SUMMED over the ranks
        # what apply() reads: a buffer nobody but all-to-all #2 fills
        self.stage_buf = torch.full((3 * B, self.d), float("nan"))
        self.stage = tuple(self.stage_buf.view(3, B, self.d)[k].numpy() for k in range(3))
        return losses, stage_slice, torch.from_numpy(self.gw)

    def stage_rows(self, B):
        return self.stage_buf


def _split_worker(rank, world, port, q, kind_name, layout):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, Q, w, wu, batches = _problem()
        kind = {"rubibceboth": oracle.LOSS_RUBIBCEBOTH, "rubibce": oracle.LOSS_RUBIBCE}[kind_name]
        model = sharded_train.RowShardedMF(torch.from_numpy(P), torch.from_numpy(Q), torch.from_numpy(w), torch.from_numpy(wu),
                                           SplitOracleBackend(kind, P.shape[1], **HYP), layout=layout)
        Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
        st = oracle.AdamState([P.shape, Q.shape, w.shape, wu.shape])
        ok, wire = True, []
        for k, (u, i, j) in enumerate(batches):
            tu, ti, tj = torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j)
            counts = model.route(tu, ti, tj)[0].tolist() if k == 1 else None          # (given by the caller once, computed inside otherwise)
            got = model.step_split(tu, ti, tj, counts=counts).numpy()
            want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, HYP["lr"], HYP["decay"], HYP["alpha"], HYP["beta"],
                                        HYP["bs"])
            ok = ok and np.allclose(got, want, rtol=1e-6)
            wire.append(model.wire_rows)
        Pf, Qf = model.full_tables()
        ok = ok and np.array_equal(Pf.numpy(), Po) and np.array_equal(Qf.numpy(), Qo)
        ok = ok and np.array_equal(model.w.numpy(), wo) and np.array_equal(model.wu.numpy(), wuo)
        B = len(batches[0][0])
        q.put((rank, bool(ok), max(wire) < 2 * 3 * B // world + 3 * B // 4))       # ~ 2 * (W-1)/W * 3B/W rows cross ranks, not 3B
    finally:
        dist.destroy_process_group()


def _run_split(world, kind_name, layout):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q, kind_name, layout)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True, True) for r in range(world)], res


def test_split_step_gloo_world2_interleaved():
    _run_split(2, "rubibceboth", "interleaved")


def test_split_step_gloo_world3_ranges_uneven_slices():
    """three ranks, contiguous row ranges (every hot item on rank 0: very unequal send counts), 96 positions in slices of 32"""
    _run_split(3, "rubibce", "range")


def test_route_tables_are_consistent():
    """counts[q][p] = rows owner q sends to slice p; every reference appears exactly once on the sending and on the receiving side"""
    P, Q, w, wu, batches = _problem()
    u, i, j = (torch.from_numpy(x) for x in batches[0])
    B, W = u.numel(), 3
    sends, recvs, tabs = [], [], []
    for r in range(W):
        m = sharded_train.RowShardedMF(torch.from_numpy(P), torch.from_numpy(Q), torch.from_numpy(w), torch.from_numpy(wu),
                                       SplitOracleBackend(1, P.shape[1], **HYP), rank=r, world=W)
        counts, send_ref, recv_ref, rows = m.route(u, i, j)
        tabs.append(counts)
        sends.append(send_ref[:int(counts[r].sum())]); recvs.append(recv_ref[:int(counts[:, r].sum())])
    assert all(torch.equal(tabs[0], t) for t in tabs) and int(tabs[0].sum()) == 3 * B
    assert sorted(torch.cat(sends).tolist()) == list(range(3 * B)) == sorted(torch.cat(recvs).tolist())
    # what q sends to p, in order, is what p expects from q, in order
    for q_ in range(W):
        off_s = 0
        for p_ in range(W):
            n = int(tabs[0][q_][p_])
            off_r = int(tabs[0][:q_, p_].sum())
            assert torch.equal(sends[q_][off_s:off_s + n], recvs[p_][off_r:off_r + n]), (q_, p_)
            off_s += n
