"""Item-catalogue sharding for multi-GPU evaluation (new functionality).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  The
catalogue is cut into `world` contiguous item ranges; every rank ranks ALL
query users against its own range (fused scoring + top-K), then the per-shard
top-K lists -- U*K*(4+4) bytes per rank, a few MB, (score, id) packed in 64-bit
words -- are exchanged with ONE all-gather and merged.  The merge is exact because the global top-K is a subset
of the union of the per-shard top-Ks, and the tie rule (score desc, id asc) is
applied identically everywhere, so 1/2/4/8-GPU results are identical.

This module is host logic only (works with gloo/CPU tensors for tests and with
nccl/HIP tensors in production); the merge kernel itself is macr_topk_merge.
"""
import os

import torch
import torch.distributed as dist


def force_collectives():
    """MACR_FORCE_COLLECTIVES=1: issue every collective even in a world of ONE rank (a test rig: one GPU can then drive the
    RCCL code paths -- all_gather_into_tensor on packed int64 device tensors, device all-reduces and broadcasts -- that
    otherwise first run on a multi-GPU node)."""
    return os.environ.get("MACR_FORCE_COLLECTIVES", "0") == "1" and dist.is_available() and dist.is_initialized()


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def item_shard_range(n_items, rank, world_size):
    """Contiguous, balanced range [lo, hi) of rank's items; shards differ by at most one item."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_topk(local_val, local_idx, group=None):
    """All-gather per-shard (U,K) lists -> (W,U,K) tensors, identical on every rank.

    ONE direct all-gather (fully connected xGMI, one hop) of U*K*8 bytes per rank: score and id travel packed in one
    64-bit word each (fp32 bits in the high half, int32 id in the low half), so the exchange is a single collective
    -- latency-bound, no bucketing or ring tuning needed at this size."""
    rank, ws = world()
    if ws == 1 and not force_collectives():
        return local_val.unsqueeze(0), local_idx.unsqueeze(0)
    if local_val.is_cuda and dist.get_backend(group) == "gloo":
        # test rig only (several ranks sharing one GPU cannot use RCCL): gloo gathers host tensors
        v, i = gather_topk(local_val.cpu(), local_idx.cpu(), group)
        return v.to(local_val.device), i.to(local_idx.device)
    packed = torch.stack([local_idx.contiguous().view(torch.int32), local_val.contiguous().view(torch.int32)], dim=-1)
    packed = packed.contiguous().view(torch.int64).squeeze(-1)                 # (U,K) int64, little endian: idx | val
    out = torch.empty((ws,) + tuple(packed.shape), dtype=torch.int64, device=packed.device)
    # output viewed as the concatenation along dim 0: the form every backend (nccl/RCCL, gloo) accepts
    dist.all_gather_into_tensor(out.view((-1,) + tuple(packed.shape[1:])), packed, group=group)
    halves = out.view(torch.int32).view((ws,) + tuple(packed.shape) + (2,))
    return halves[..., 1].contiguous().view(torch.float32), halves[..., 0].contiguous()


def max_over_ranks(x, device):
    """max of a python float over ranks (bench timing contract)."""
    rank, ws = world()
    if ws == 1 and not force_collectives():
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def is_main():
    """True on rank 0 (and without a process group): the rank that prints, logs and writes checkpoints."""
    return world()[0] == 0


def broadcast_params(tensors, src=0, group=None):
    """Make every rank hold rank `src`'s copy of the model before an item-sharded evaluation: the ranks of a CLI run
    train replicas on identical batches, but floating-point atomics add in a different order on every GPU, so the
    replicas drift apart bit by bit -- and a sharded ranking must score ONE model."""
    if world()[1] == 1 and not force_collectives():
        return
    for t in tensors:
        if t.is_cuda and dist.get_backend(group) == "gloo":     # test rig: several ranks on one GPU
            h = t.cpu()
            dist.broadcast(h, src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src, group=group)
