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
        for role, idx, tab, lo, hi in ((0, u, P, shard.u_lo, shard.u_hi), (1, i, Q, shard.i_lo, shard.i_hi),
                                       (2, j, Q, shard.i_lo, shard.i_hi)):
            idx = idx.numpy()
            own = (idx >= lo) & (idx < hi)
            rows3[role, own] = tab[idx[own] - lo]
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
        return losses, torch.from_numpy(self.gw)

    def apply(self, shard, u, i, j):
        o = self.o
        lr_t = o.lib().orc_adam_lr_t(self.lr, self.power)
        gP, gQ = np.zeros_like(shard.P.numpy()), np.zeros_like(shard.Q.numpy())
        for grad, idx, g, lo, hi in ((self.stage[0], u, gP, shard.u_lo, shard.u_hi), (self.stage[1], i, gQ, shard.i_lo, shard.i_hi),
                                     (self.stage[2], j, gQ, shard.i_lo, shard.i_hi)):
            idx = idx.numpy()
            for t in range(len(idx)):                 # batch order, like orc_scatter_add_rows
                if lo <= idx[t] < hi:
                    g[idx[t] - lo] += grad[t]
        adam = lambda th, m, v, gr: o.lib().orc_adam_dense(th.numpy(), m.numpy(), v.numpy(), _vp(np.ascontiguousarray(gr)),
                                                           th.numel(), lr_t, 0.9, 0.999, 1e-8)
        adam(shard.P, shard.mP, shard.vP, gP)
        adam(shard.Q, shard.mQ, shard.vQ, gQ)
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


def _worker(rank, world, port, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, Q, w, wu, batches = _problem()
        kind = oracle.LOSS_RUBIBCEBOTH
        model = sharded_train.RowShardedMF(torch.from_numpy(P), torch.from_numpy(Q), torch.from_numpy(w), torch.from_numpy(wu),
                                           OracleBackend(kind, P.shape[1], **HYP))
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
        ok = ok and model.P.shape[0] == sharded_train.row_range(P.shape[0], rank, world)[1] - sharded_train.row_range(P.shape[0], rank, world)[0]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_row_sharded_training_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
