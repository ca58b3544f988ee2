"""Routing of the row-sharded split step on one GPU, as rank 0 of a simulated world of W: RowShardedMF.route() by the ~10 torch
launches it used to be (MACR_SHARD_ROUTE_TORCH=1) against the one macr_shard_route launch -- wall time per call with the stream
drained, configs[4] batch (B = 8192; also 65 536).  No collectives involved: route() is a function of the batch alone.
python tools/bench_shard_route.py [W]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from macr_amd import ops, sharded_train

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
d, n_users, n_items = 128, 1_250_000, 125_000
out = {"world": W, "layout": "interleaved"}
for B in (8192, 65536):
    hyper = ops.make_hyper(1e-3, 1e-5, 1e-3, 1e-3, B)
    gen = torch.Generator(device=dev).manual_seed(5)
    shards = ((torch.randn((n_users // W, d), generator=gen, device=dev) * 0.05), (torch.randn((n_items // W, d), generator=gen, device=dev) * 0.05),
              n_users, n_items)
    w = torch.randn(d, generator=gen, device=dev) * 0.2; wu = torch.randn(d, generator=gen, device=dev) * 0.2
    m = sharded_train.RowShardedMF(None, None, w, wu, sharded_train.HipBackend(ops.LOSS_RUBIBCEBOTH, d, hyper, dev), rank=0, world=W, shards=shards)
    rs = np.random.RandomState(1)
    u = torch.from_numpy(rs.choice(n_users, B, replace=False).astype(np.int32)).to(dev)
    i = torch.from_numpy((rs.zipf(1.2, B) % n_items).astype(np.int32)).to(dev)
    j = torch.from_numpy(rs.randint(0, n_items, B).astype(np.int32)).to(dev)
    res = {}
    ref = None
    for mode in ("torch", "kernel"):
        os.environ["MACR_SHARD_ROUTE_TORCH"] = "1" if mode == "torch" else "0"
        for _ in range(5):
            r = m.route(u, i, j)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            r = m.route(u, i, j)
        torch.cuda.synchronize()
        res[mode + "_us"] = 1e6 * (time.perf_counter() - t0) / 50
        got = (r[0].cpu().numpy(), r[1].cpu().numpy(), r[2].cpu().numpy())
        if ref is None:
            ref = got
        else:
            n_send, n_recv = int(got[0][0].sum()), int(got[0][:, 0].sum())
            res["equal"] = bool(np.array_equal(ref[0], got[0]) and np.array_equal(ref[1][:n_send], got[1][:n_send])
                                and np.array_equal(ref[2][:n_recv], got[2][:n_recv]))
    out["B=%d" % B] = res
print(json.dumps(out))
