"""Worker of tests/test_gpu_product.py::test_row_sharded_training_two_ranks_one_gpu: each rank (torch.distributed.run,
gloo rig: both ranks on GPU 0) trains its row shard with macr_amd.sharded_train on the HIP backend, rank 0 compares the
reassembled model and the losses with the single-GPU step (ops.MFState) and with the CPU oracle, and prints a verdict."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import oracle
    from macr_amd import ops, sharded_train
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(21)
    n_users, n_items, d, B = 5003, 1201, 64, 1500           # B not a multiple of 256, shards of unequal size
    P = (rs.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    Q = (rs.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    w, wu = (rs.standard_normal(d) * 0.3).astype(np.float32), (rs.standard_normal(d) * 0.3).astype(np.float32)
    lr, decay, alpha, beta, bs = 1e-3, 1e-5, 1e-2, 1e-3, 1024
    kind = ops.LOSS_RUBIBCEBOTH
    t = lambda a: torch.from_numpy(a).to(dev)
    hyper = ops.make_hyper(lr, decay, alpha, beta, bs)
    model = sharded_train.RowShardedMF(t(P), t(Q), t(w), t(wu), sharded_train.HipBackend(kind, d, hyper, dev))
    split = os.environ.get("MACR_SHARD_SPLIT", "0") == "1"          # forward / backward split over the ranks, all-to-all exchange
    single = ops.MFState(t(P), t(Q), t(w), t(wu), hyper, B)
    st = oracle.AdamState([P.shape, Q.shape, (d,), (d,)])
    Po, Qo, wo, wuo = P.copy(), Q.copy(), w.copy(), wu.copy()
    worst = 0.0
    for step in range(int(os.environ.get("MACR_TEST_STEPS", "3"))):
        u = rs.choice(n_users, B, replace=False).astype(np.int32)
        i = (rs.zipf(1.3, B) % n_items).astype(np.int32)
        j = rs.randint(0, n_items, B).astype(np.int32)
        model.split = split
        got = model.step(t(u), t(i), t(j)).cpu().numpy()
        ref = single.step(kind, t(u), t(i), t(j)).cpu().numpy()
        want = oracle.mf_train_step(kind, u, i, j, Po, Qo, wo, wuo, st, lr, decay, alpha, beta, bs)
        worst = max(worst, float(np.abs(got / want - 1).max()), float(np.abs(got / ref - 1).max()))
    Pf, Qf = model.full_tables()
    ok_loss = worst < 1e-5
    # same tolerances as the single-GPU train-step tests (tests/test_gpu_ops.py): 0.2 % of an Adam step
    close = lambda a, b: float(np.abs(a - b).max())
    dP, dQ = close(Pf.cpu().numpy(), Po), close(Qf.cpu().numpy(), Qo)
    dPs, dQs = close(Pf.cpu().numpy(), single.P.cpu().numpy()), close(Qf.cpu().numpy(), single.Q.cpu().numpy())
    dw = max(close(model.w.cpu().numpy(), wo), close(model.wu.cpu().numpy(), wuo))
    tol = 2e-3 * lr * int(os.environ.get("MACR_TEST_STEPS", "3"))
    # w, w_user must be bit-identical on every rank (branch-vector gradients are broadcast from rank 0)
    both = [torch.zeros(2 * d, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(both, torch.cat([model.w, model.wu]).cpu())
    same_w = all(torch.equal(both[0], b) for b in both)
    shard_rows = model.P.shape[0] + model.Q.shape[0]
    if rank == 0:
        print(json.dumps({"ok": bool(ok_loss and max(dP, dQ, dPs, dQs, dw) < tol and same_w), "worst_loss_rel": worst,
                          "dP": dP, "dQ": dQ, "dP_single": dPs, "dQ_single": dQs, "dw": dw, "tol": tol, "same_w": same_w,
                          "world": world, "rows_on_rank0": shard_rows, "rows_total": n_users + n_items, "split": split,
                          "wire_rows": getattr(model, "wire_rows", None), "batch_rows": 3 * B, "lazy_period": model.lazy_period}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
