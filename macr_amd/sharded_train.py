"""Row-sharded MF training over several GPUs (new functionality; SURVEY.md 8e, BASELINE configs[4]).

The reference keeps each embedding table in one tf.Variable (macr_mf/model.py:112-113).  At 10 M users x 1 M items,
d = 128, TF-style dense Adam streams 24*d bytes of EVERY row per step -- 33.8 GB -- which is what a step costs; the
batch itself is a few MB.  So the rows of P and Q (with their Adam slots and gradient scratch) are range-sharded over
the ranks, one process per GPU, and a step exchanges only batch-sized data:

    1. gather     each rank writes the batch rows it owns into a zero (3,B,d) buffer; ONE all-reduce(sum) makes the
                  batch's 3B rows resident everywhere (every row has exactly one owner)         3*B*d*4 bytes
    2. forward    per-pair dots / branch factors of the whole batch, redundantly on every rank (deterministic kernel,
                  so p, n, a, b need no exchange)
    3. (B,B)      rank r evaluates its share of the row blocks; ONE all-reduce(sum) of the partial row/column sums
                  and loss partials                                                             (nrb+ncb)*2*B*4 bytes
    4. backward   gradient rows of the whole batch, redundantly (a few MB of HBM traffic)
                  + broadcast of rank 0's branch-vector partials (2*8*d floats) so that w, w_user never drift
    5. apply      every rank segment-reduces the references to ITS rows and runs dense Adam on ITS shard: the 24*d*rows
                  bytes per step are divided by the number of ranks

All three collectives are latency-class messages on xGMI (12.6 MB + 0.5 MB + 8 KB at B = 8192, d = 128).  The math is
the single-GPU step's (macr_mf_train_step) up to summation order; `backend` is the device half (HIP: the macr_shard_*
entry points of include/macr_hip.h; the CPU tests plug the oracle in its place)."""
import ctypes

import torch
import torch.distributed as dist

from . import sharding


def row_range(n_rows, rank, world):
    """Contiguous, balanced [lo, hi) of the rows rank owns (same rule as the evaluator's item shards)."""
    return sharding.item_shard_range(n_rows, rank, world)


class HipBackend(object):
    """Device half on the MI355X: thin calls into libmacr_hip.so (no fallback)."""

    def __init__(self, kind, d, hyper, device):
        from . import _lib, ops
        self._lib, self.ops, self.kind, self.d, self.hyper, self.device = _lib, ops, kind, d, hyper, device
        self.ws, self.cap = None, 0
        self.adam_pow = torch.tensor([hyper.beta1, hyper.beta2], dtype=torch.float32, device=device)
        self.losses = torch.zeros(3, dtype=torch.float32, device=device)

    def _reserve(self, B):
        if B > self.cap:
            n = self._lib.lib().macr_shard_workspace_bytes(B, self.d)
            if n == 0:
                raise self._lib.MacrError(self._lib.E_UNSUPPORTED, "embed_size %d not in {32,64,128,256}" % self.d)
            self.ws, self.cap = torch.empty(n, dtype=torch.uint8, device=self.device), B
        if getattr(self, "rows3", None) is None or self.rows3.shape[1] != B:
            self.rows3 = torch.empty((3, B, self.d), dtype=torch.float32, device=self.device)

    def _view(self, ptr, nbytes):
        """fp32 tensor over a region of the workspace the library pointed at"""
        off = ptr.value - self.ws.data_ptr()
        return self.ws[off:off + nbytes.value].view(torch.float32)

    def gather(self, shard, u, i, j):
        B = u.numel()
        self._reserve(B)
        o, L = self.ops, self._lib.lib()
        self._lib.check(L.macr_shard_gather(B, self.d, o._ptr(shard.P), shard.u_lo, shard.u_hi, o._ptr(shard.Q), shard.i_lo,
                                            shard.i_hi, o._ptr(u), o._ptr(i), o._ptr(j), o._ptr(self.rows3), o._stream()))
        return self.rows3

    def forward_and_bxb(self, shard, rows3, rank, world):
        o, L = self.ops, self._lib.lib()
        B = rows3.shape[1]
        self._lib.check(L.macr_shard_forward(self.kind, B, self.d, o._ptr(rows3), o._ptr(shard.w), o._ptr(shard.wu),
                                             o._ptr(self.ws), self.ws.numel(), o._stream()))
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._lib.check(L.macr_shard_bxb(B, self.d, rank, world, ctypes.byref(ptr), ctypes.byref(nbytes), o._ptr(self.ws),
                                         self.ws.numel(), o._stream()))
        return self._view(ptr, nbytes)                      # to be summed over the ranks

    def backward(self, shard, rows3):
        o, L = self.ops, self._lib.lib()
        B = rows3.shape[1]
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._lib.check(L.macr_shard_backward(self.kind, B, self.d, o._ptr(rows3), o._ptr(shard.w), o._ptr(shard.wu),
                                              o._ptr(self.adam_pow), ctypes.byref(self.hyper), o._ptr(self.losses),
                                              ctypes.byref(ptr), ctypes.byref(nbytes), o._ptr(self.ws), self.ws.numel(),
                                              o._stream()))
        return self.losses, self._view(ptr, nbytes)         # losses; branch-vector partial rows (broadcast from rank 0)

    def apply(self, shard, u, i, j):
        o, L = self.ops, self._lib.lib()
        self._lib.check(L.macr_shard_apply(self.kind, u.numel(), self.d, shard.u_hi - shard.u_lo, shard.i_hi - shard.i_lo,
                                           shard.u_lo, shard.i_lo, o._ptr(u), o._ptr(i), o._ptr(j),
                                           o._ptr(shard.P), o._ptr(shard.Q), o._ptr(shard.w), o._ptr(shard.wu),
                                           o._ptr(shard.mP), o._ptr(shard.vP), o._ptr(shard.mQ), o._ptr(shard.vQ),
                                           o._ptr(shard.mw), o._ptr(shard.vw), o._ptr(shard.mwu), o._ptr(shard.vwu),
                                           o._ptr(shard.gP), o._ptr(shard.gQ), o._ptr(shard.tP), o._ptr(shard.tQ),
                                           ctypes.byref(self.hyper), o._ptr(self.ws), self.ws.numel(), o._stream()))


class RowShardedMF(object):
    """This rank's shard of the MF model + its optimizer state.  P_full / Q_full (any rank-identical source) are only
    sliced at construction; afterwards a rank holds rows [u_lo,u_hi) of P and [i_lo,i_hi) of Q, nothing else."""

    def __init__(self, P_full, Q_full, w, wu, backend, rank=None, world=None, group=None, shards=None):
        """shards=(P_shard, Q_shard, n_users, n_items): this rank's rows directly (P_full / Q_full are ignored) -- for
        tables whose full copy exists nowhere (10 M x 1 M rows, d = 128)."""
        r, ws = sharding.world()
        self.rank, self.world, self.group = (r if rank is None else rank), (ws if world is None else world), group
        self.n_users, self.n_items = (shards[2], shards[3]) if shards else (P_full.shape[0], Q_full.shape[0])
        self.u_lo, self.u_hi = row_range(self.n_users, self.rank, self.world)
        self.i_lo, self.i_hi = row_range(self.n_items, self.rank, self.world)
        clone = lambda t: t.clone().contiguous()
        if shards:
            self.P, self.Q = shards[0].contiguous(), shards[1].contiguous()
            assert self.P.shape[0] == self.u_hi - self.u_lo and self.Q.shape[0] == self.i_hi - self.i_lo
        else:
            self.P, self.Q = clone(P_full[self.u_lo:self.u_hi]), clone(Q_full[self.i_lo:self.i_hi])
        self.w, self.wu = clone(w.reshape(-1)), clone(wu.reshape(-1))
        self.collective_ms = None                   # bench: {"rows": [...], "partials": [...], "branch": [...]} event times
        z = torch.zeros_like
        self.mP, self.vP, self.mQ, self.vQ = z(self.P), z(self.P), z(self.Q), z(self.Q)
        self.mw, self.vw, self.mwu, self.vwu = z(self.w), z(self.w), z(self.wu), z(self.wu)
        self.gP, self.gQ = z(self.P), z(self.Q)
        self.tP = torch.zeros(self.P.shape[0], dtype=torch.int32, device=self.P.device)
        self.tQ = torch.zeros(self.Q.shape[0], dtype=torch.int32, device=self.Q.device)
        self.backend = backend

    # ------------------------------------------------------------------ collectives (RCCL over xGMI; gloo in the tests)
    def _host_rig(self, t):
        return t.is_cuda and self.world > 1 and dist.get_backend(self.group) == "gloo"

    def _timed(self, name, fn):
        """run a collective; with collective_ms set, bracket it with events on the current stream (read them after a sync)"""
        if self.collective_ms is None or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        self.collective_ms.setdefault(name, []).append((e0, e1))

    def _all_reduce(self, t, name="all_reduce"):
        if self.world == 1 and not sharding.force_collectives():
            return
        if self._host_rig(t):                       # test rig: several ranks on one GPU cannot use RCCL
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self._timed(name, lambda: dist.all_reduce(t, group=self.group))

    def _broadcast(self, t, src=0, name="broadcast"):
        if self.world == 1 and not sharding.force_collectives():
            return
        if self._host_rig(t):
            h = t.cpu()
            dist.broadcast(h, src, group=self.group)
            t.copy_(h)
        else:
            self._timed(name, lambda: dist.broadcast(t, src, group=self.group))

    # ------------------------------------------------------------------ one step
    def step(self, u, i, j):
        """u, i, j: the SAME batch on every rank (int32, global row ids).  Returns {loss, mf_loss, reg_loss} (3,)."""
        be = self.backend
        rows3 = be.gather(self, u, i, j)
        self._all_reduce(rows3, "rows")                           # 1. the batch's rows, everywhere
        partials = be.forward_and_bxb(self, rows3, self.rank, self.world)
        if partials is not None:
            self._all_reduce(partials, "partials")                # 3. row / column sums of the (B,B) term
        losses, branch = be.backward(self, rows3)
        if branch is not None:
            self._broadcast(branch, 0, "branch")                  # 4. one copy of the branch-vector gradients
        be.apply(self, u, i, j)                                   # 5. local segment reduce + dense Adam on the shard
        return losses

    def full_tables(self):
        """(P, Q) reassembled on every rank (tests / checkpoints; not on the training path)."""
        def cat(local, n_rows):
            if self.world == 1:
                return local.clone()
            d = local.shape[1]
            full = torch.zeros((n_rows, d), dtype=local.dtype, device=local.device)
            lo, _ = row_range(n_rows, self.rank, self.world)
            full[lo:lo + local.shape[0]] = local
            self._all_reduce(full)
            return full
        return cat(self.P, self.n_users), cat(self.Q, self.n_items)
