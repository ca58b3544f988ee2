"""N>1 host logic on CPU: item sharding + the all-gather of per-shard top-K over gloo (world_size 2).
The oracle plays the device kernels here (it is the checker, allowed in tests): each rank ranks its own
item shard, the product's gather_topk exchanges the lists, and the merged result must equal the
unsharded ranking bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from macr_amd import sharding


def test_item_shard_ranges_partition_the_catalogue():
    for n, w in ((744, 1), (744, 2), (40981, 8), (7, 8), (8790, 4), (1000003, 8)):
        prev = 0
        sizes = []
        for r in range(w):
            lo, hi = sharding.item_shard_range(n, r, w)
            assert lo == prev and hi >= lo
            sizes.append(hi - lo)
            prev = hi
        assert prev == n and max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(3)                      # same data on every rank
        U, N, d, K = 37, 501, 16, 20
        P = rs.standard_normal((U, d)).astype(np.float32)
        Q = rs.standard_normal((N, d)).astype(np.float32)
        Q[10:20] = Q[300:310]                              # exact ties across shards
        mask = [sorted(rs.choice(N, 9, replace=False).tolist()) for _ in range(U)]
        mask[0] = list(range(0, N - 3))                    # fewer than K candidates
        csr = oracle.csr_from_lists(mask)
        lo, hi = sharding.item_shard_range(N, rank, world)
        v, i, _ = oracle.score_topk(oracle.SCORE_NORMAL, P, Q[lo:hi], K, mask=csr, item_offset=lo)
        gv, gi = sharding.gather_topk(torch.from_numpy(v), torch.from_numpy(i))
        assert gv.shape == (world, U, K)
        mv, mi, mc = oracle.topk_merge(gv.numpy(), gi.numpy())
        wv, wi, wc = oracle.score_topk(oracle.SCORE_NORMAL, P, Q, K, mask=csr)
        ok = np.array_equal(mi, wi) and np.array_equal(mv, wv) and np.array_equal(mc, wc)
        t = sharding.max_over_ranks(float(rank + 1), torch.device("cpu"))
        q.put((rank, bool(ok), t))
    finally:
        dist.destroy_process_group()


def test_sharded_topk_all_gather_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]               # max over ranks reached everybody
