"""Row-sharded training (macr_shard_*, macr_amd/sharded_train.py) at ONE rank's share of BASELINE configs[4]
(10 M users x 1 M items, d = 128, 8 ranks): 1 250 000 user rows + 125 000 item rows, B = 8192, rubibceboth.
world = 1 here (the three collectives are no-ops): the per-call kernel times of a step and the step time of the
rank-local work; the collectives' bytes are stated beside them (not measured: no multi-GPU box).
python tools/bench_shard_train.py [n_users n_items d B]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from macr_amd import ops, sharded_train

n_users, n_items, d, B = ([int(x) for x in sys.argv[1:5]] + [1_250_000, 125_000, 128, 8192][len(sys.argv) - 1:])[:4]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(5)
P = (torch.randn((n_users, d), generator=gen, device=dev) * 0.05).contiguous()
Q = (torch.randn((n_items, d), generator=gen, device=dev) * 0.05).contiguous()
w = torch.randn(d, generator=gen, device=dev) * 0.2; wu = torch.randn(d, generator=gen, device=dev) * 0.2
hyper = ops.make_hyper(1e-3, 1e-5, 1e-3, 1e-3, B)
model = sharded_train.RowShardedMF(P, Q, w, wu, sharded_train.HipBackend(ops.LOSS_RUBIBCEBOTH, d, hyper, dev), rank=0, world=1)
rs = np.random.RandomState(1)
batches = []
for _ in range(8):
    u = torch.from_numpy(rs.choice(n_users, B, replace=False).astype(np.int32)).to(dev)
    i = torch.from_numpy((rs.zipf(1.2, B) % n_items).astype(np.int32)).to(dev)
    j = torch.from_numpy(rs.randint(0, n_items, B).astype(np.int32)).to(dev)
    batches.append((u, i, j))
for k in range(3):
    model.step(*batches[k])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for k in range(n):
    model.step(*batches[k % 8])
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / n
ops.timing_begin()
for k in range(4):
    model.step(*batches[k])
marks = ops.timing_end(1024)
ku = {}
for name, t in marks:
    ku[name] = ku.get(name, 0.0) + 1e3 * t / 4
rows = n_users + n_items
adam_bytes = 24.0 * d * rows
out = {"workload": "configs[4], one rank of 8: %d + %d rows, d=%d, B=%d, rubibceboth, world=1" % (n_users, n_items, d, B),
       "ms_per_step_rank_local": ms, "kernels_us_per_step": {k: round(v, 1) for k, v in ku.items()},
       "adam_pass_bytes": adam_bytes, "adam_pass_at_step_time_GBps": adam_bytes / (ms * 1e-3) / 1e9,
       "collectives_per_step_bytes": {"all_reduce_rows3": 3 * B * d * 4, "all_reduce_bxb_partials": "~0.5 MB",
                                      "broadcast_branch_partials": 8 * 2 * d * 4},
       "note": "rank-local compute only; the three collectives are batch-sized (12.6 MB + 0.5 MB + 8 KB) and unmeasured"}
print(json.dumps(out))
