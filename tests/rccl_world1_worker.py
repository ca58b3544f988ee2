"""Worker of tests/test_gpu_product.py::test_rccl_code_paths_in_a_world_of_one: a process group of ONE rank on backend
"nccl" (= RCCL) with MACR_FORCE_COLLECTIVES=1, so that every collective of the multi-GPU paths is really issued on device
tensors -- the packed 64-bit all_gather_into_tensor of the evaluator, the device all-reduces and the broadcast of the
row-sharded trainer -- instead of first running on whatever multi-GPU node appears.  Prints a JSON verdict."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ["MACR_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from macr_amd import ops, sharding, sharded_train
    from macr_amd.evaluator import Evaluator
    assert dist.get_backend() == "nccl" and sharding.force_collectives()
    res = {}
    rs = np.random.RandomState(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # 1. the evaluator's exchange: packed (score, id) all-gather, bit-exact round trip incl. -inf, -1 and negative zero
    val = rs.standard_normal((300, 20)).astype(np.float32); val[0, :3] = [-np.inf, -0.0, 0.0]
    idx = rs.randint(0, 1 << 30, (300, 20)).astype(np.int32); idx[1, :2] = -1
    gv, gi = sharding.gather_topk(t(val), t(idx))
    res["gather_topk"] = bool(gv.shape == (1, 300, 20) and np.array_equal(gv[0].cpu().numpy().view(np.uint32), val.view(np.uint32))
                              and np.array_equal(gi[0].cpu().numpy(), idx))
    res["max_over_ranks"] = sharding.max_over_ranks(1.25, dev) == 1.25
    x = t(rs.standard_normal(1000).astype(np.float32)); x0 = x.clone()
    sharding.broadcast_params([x])
    res["broadcast_params"] = bool(torch.equal(x, x0))
    # 2. a full sharded evaluation through the two-graph path would need world > 1; the collective itself is covered above.
    # 3. the row-sharded trainer with its three collectives issued (all-reduce rows, all-reduce partials, broadcast branch)
    n_users, n_items, d, B = 3001, 901, 64, 700
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32); Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    hyper = ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, 1024)
    kind = ops.LOSS_RUBIBCEBOTH
    model = sharded_train.RowShardedMF(t(P), t(Q), t(w), t(wu), sharded_train.HipBackend(kind, d, hyper, dev))
    model.collective_ms = {}
    single = ops.MFState(t(P), t(Q), t(w), t(wu), hyper, B)
    worst = 0.0
    for step in range(3):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.3, B) % n_items).astype(np.int32); j = rs.randint(0, n_items, B).astype(np.int32)
        a = model.step(t(u), t(i), t(j)).cpu().numpy()
        b = single.step(kind, t(u), t(i), t(j)).cpu().numpy()
        worst = max(worst, float(np.abs(a - b).max() / np.abs(b).max()))
    torch.cuda.synchronize()
    res["collectives_issued"] = {k: len(v) for k, v in model.collective_ms.items()}
    res["sharded_vs_single_loss_rel"] = worst
    Pf, Qf = model.full_tables()
    res["tables_max_diff"] = float(max((Pf - single.P).abs().max(), (Qf - single.Q).abs().max()))
    res["ok"] = bool(res["gather_topk"] and res["max_over_ranks"] and res["broadcast_params"] and worst < 1e-5
                     and res["tables_max_diff"] < 2e-3 * 1e-3 * 3
                     and res["collectives_issued"] == {"rows": 3, "partials": 3, "branch": 3})
    print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
