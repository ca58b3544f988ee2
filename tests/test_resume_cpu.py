"""--resume on several ranks, host logic on CPU (gloo, world size 2): macr_amd.train_state.resume must read files on the
main rank only, must never call the model's (collective, for a row-sharded model) state_dict(), and must leave every rank
with ITS rows of the checkpoint -- the round-4 version reassembled a model that only rank 0 had and deadlocked."""
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from macr_amd import sharded_train, train_state


class _ShardedToy(object):
    """What train_state needs of mf.ShardedBPRMF: a device, load_state_dict(full tables) taking the owned rows."""
    sharded = True

    def __init__(self, n, d, rank, world):
        self.device = torch.device("cpu")
        self.own = sharded_train.Owned(n, rank, world)
        self.P = torch.zeros((self.own.n, d))
        self.m = torch.zeros((self.own.n, d))
        self.w = torch.zeros(d)
        self.rubi_c = -1.0
        self.pow = torch.zeros(2)

    def state_dict(self):
        raise AssertionError("resume() must not go through the collective state_dict()")

    def load_state_dict(self, sd):
        self.P.copy_(self.own.take(sd["user_embedding"])); self.m.copy_(self.own.take(sd["opt1.mP"]))
        self.w.copy_(sd["w"]); self.rubi_c = float(sd["rubi_c"]); self.pow.copy_(sd["opt1.adam_pow"])


def _full(n, d):
    rs = np.random.RandomState(5)
    return {"user_embedding": torch.from_numpy(rs.standard_normal((n, d)).astype(np.float32)),
            "opt1.mP": torch.from_numpy(rs.standard_normal((n, d)).astype(np.float32)),
            "w": torch.from_numpy(rs.standard_normal(d).astype(np.float32)), "rubi_c": 37.5,
            "opt1.adam_pow": torch.tensor([0.81, 0.998]), "row_shard_kind": 1}


def _worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, d = 101, 8
        model = _ShardedToy(n, d, rank, world)

        def latest():
            assert rank == 0, "only the main rank looks for checkpoints"
            return 7, os.path.join(tmp, "7_ckpt.pt")
        import random
        random.seed(99 + rank); np.random.seed(99 + rank)
        epoch, book = train_state.resume(model, latest, lambda e: os.path.join(tmp, "%d_train_state.json" % e))
        full = _full(n, d)
        ok = epoch == 7 and book == {"stopping_step": 3, "config": {"best_hr": 0.25}}
        ok = ok and torch.equal(model.P, model.own.take(full["user_embedding"])) and torch.equal(model.m, model.own.take(full["opt1.mP"]))
        ok = ok and torch.equal(model.w, full["w"]) and model.rubi_c == 37.5 and torch.equal(model.pow, full["opt1.adam_pow"])
        ok = ok and model.P.shape[0] in (50, 51)
        q.put((rank, bool(ok), random.random(), float(np.random.rand())))      # the host RNG streams continue identically
    finally:
        dist.destroy_process_group()


def test_resume_row_sharded_world2(tmp_path):
    import random
    torch.save(_full(101, 8), tmp_path / "7_ckpt.pt")
    random.seed(1234); np.random.seed(4321)
    train_state.save(str(tmp_path / "7_train_state.json"), {"stopping_step": 3, "config": {"best_hr": 0.25}})
    want = (random.random(), float(np.random.rand()))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(60)
    assert [(r[0], r[1]) for r in res] == [(0, True), (1, True)]
    assert all((r[2], r[3]) == want for r in res)


def test_resume_without_side_file_warns(tmp_path):
    """weights with no usable JSON next to them resume the model only -- and say so (round 4 dropped the bookkeeping silently)"""
    torch.save(_full(11, 4), tmp_path / "3_ckpt.pt")
    said = []
    model = _ShardedToy(11, 4, 0, 1)
    epoch, book = train_state.resume(model, lambda: (3, str(tmp_path / "3_ckpt.pt")),
                                     lambda e: str(tmp_path / ("%d_train_state.json" % e)), warn=said.append)
    assert epoch == 3 and book is None and len(said) == 1 and "weights only" in said[0]
    with open(tmp_path / "3_train_state.json", "w") as f:
        json.dump({"format": 2}, f)
    said.clear()
    epoch, book = train_state.resume(model, lambda: (3, str(tmp_path / "3_ckpt.pt")),
                                     lambda e: str(tmp_path / ("%d_train_state.json" % e)), warn=said.append)
    assert epoch == 3 and book is None and "format 2" in said[0]
    assert model.rubi_c == 37.5
