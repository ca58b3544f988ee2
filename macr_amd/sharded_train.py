"""Row-sharded MF training over several GPUs (new functionality; SURVEY.md 8e, BASELINE configs[4]).

The reference keeps each embedding table in one tf.Variable (macr_mf/model.py:112-113).  At 10 M users x 1 M items,
d = 128, TF-style dense Adam streams 24*d bytes of EVERY row per step -- 33.8 GB -- which is what a step costs; the
batch itself is a few MB.  So the rows of P and Q (with their Adam slots and gradient scratch) are sharded over the ranks
(interleaved: row r on rank r % W, see Owned), one process per GPU, and a step exchanges only batch-sized data:

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
the single-GPU step's (macr_mf_train_step) up to summation order, for all three loss kinds (`normalbce` has no (B,B)
term and no branch vectors: steps 2-4 are one kernel and there is ONE collective per step); `backend` is the device half (HIP: the macr_shard_*
entry points of include/macr_hip.h; the CPU tests plug the oracle in its place)."""
import ctypes
import os

import torch
import torch.distributed as dist

from . import sharding


def row_range(n_rows, rank, world):
    """Contiguous, balanced [lo, hi) of the rows rank owns (same rule as the evaluator's item shards)."""
    return sharding.item_shard_range(n_rows, rank, world)


class Owned(object):
    """Which rows of a table a rank holds: local row l is global row lo + l * stride, l < n.
    "interleaved" (default): row r lives on rank r % world -- item ids follow popularity in recommender data, so a
    contiguous range puts every hot item (and its Adam traffic, its gradient references) on rank 0; interleaved, the hot
    rows and the row counts spread evenly (they differ by at most one row).  "range": contiguous, balanced ranges."""

    def __init__(self, n_rows, rank, world, layout="interleaved"):
        self.n_rows, self.rank, self.world, self.layout = n_rows, rank, world, layout
        if layout == "interleaved":
            self.lo, self.stride = rank, world
            self.n = (n_rows - rank + world - 1) // world if n_rows > rank else 0
        elif layout == "range":
            lo, hi = row_range(n_rows, rank, world)
            self.lo, self.stride, self.n = lo, 1, hi - lo
        else:
            raise ValueError("layout must be 'interleaved' or 'range'")

    def global_ids(self, device=None):
        return torch.arange(self.n, dtype=torch.long, device=device) * self.stride + self.lo

    def take(self, full):
        """this rank's rows of a full table"""
        return full[self.lo::self.stride][:self.n] if self.stride > 1 else full[self.lo:self.lo + self.n]

    def owner_of(self, ids):
        """rank that owns each of the global row ids (any rank can tell: the layout is a function of the id)"""
        if self.layout == "interleaved":
            return ids % self.world
        cache = self.__dict__.setdefault("_bounds", {})               # per device: no host-to-device copy per step
        bounds = cache.get(ids.device)
        if bounds is None:
            bounds = cache[ids.device] = torch.tensor([row_range(self.n_rows, r, self.world)[1] for r in range(self.world)],
                                                      device=ids.device)
        return torch.bucketize(ids, bounds, right=True)

    def local_index(self, ids):
        """local row of ids THIS rank owns"""
        return (ids - self.lo) // self.stride

    def local_of(self, ids):
        """(mask of the ids this rank owns, their local indices) for an integer tensor / array of global row ids"""
        rel = ids - self.lo
        own = (rel >= 0) & (rel % self.stride == 0) & (rel // self.stride < self.n)
        return own, rel // self.stride


def lazy_period_for(n_rows, d, target_bytes=256 << 20):
    """K of the lazy dense Adam pass for a shard of n_rows rows: the dense pass moves 24*d bytes per row and step; sweep a K-th
    of that per step, about target_bytes (40 us of HBM time: below the VALU time of the pass's arithmetic, which no period
    removes -- measured at 11 M rows, d = 128: 6.36 ms per step dense, 1.04 / 0.66 / 0.68 / 0.67 ms with K = 8 / 16 / 32 / 63,
    profiles/r05_c4_lazy_steady.txt) -- 1 (dense every step) for tables the size of the reference's datasets"""
    k = int(round(24.0 * d * n_rows / target_bytes))
    return max(1, min(k, 64))


class HipBackend(object):
    """Device half on the MI355X: thin calls into libmacr_hip.so (no fallback)."""
    lazy_capable = True                 # the lazy dense Adam pass (include/macr_hip.h: macr_lazy_adam) exists on this backend

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

    def _lazy(self, shard):
        """struct macr_lazy_adam of a shard whose dense pass is blocked in time (None: dense every step)"""
        if shard.lazy_period <= 1:
            return None
        o = self.ops
        return self._lib.LazyAdam(o._ptr(shard.lazy_state), o._ptr(shard.stP), o._ptr(shard.stQ), shard.lazy_period)

    def gather(self, shard, u, i, j):
        B = u.numel()
        self._reserve(B)
        o, L = self.ops, self._lib.lib()
        ou, oi = shard.own_u, shard.own_i
        lz = self._lazy(shard)
        if lz is not None:
            self._lib.check(L.macr_shard_gather_lazy(B, self.d, o._ptr(shard._P), o._ptr(shard._mP), o._ptr(shard._vP), ou.lo, ou.stride,
                                                     ou.n, o._ptr(shard._Q), o._ptr(shard._mQ), o._ptr(shard._vQ), oi.lo, oi.stride, oi.n,
                                                     o._ptr(u), o._ptr(i), o._ptr(j), ctypes.byref(self.hyper), ctypes.byref(lz),
                                                     o._ptr(self.rows3), o._stream()))
            return self.rows3
        self._lib.check(L.macr_shard_gather(B, self.d, o._ptr(shard._P), ou.lo, ou.stride, ou.n, o._ptr(shard._Q), oi.lo, oi.stride,
                                            oi.n, o._ptr(u), o._ptr(i), o._ptr(j), o._ptr(self.rows3), o._stream()))
        return self.rows3

    def lazy_rows(self, shard, table, rows):
        """(n, d) rows of the local table `table` ("P" / "Q") as of the current step; rows: int32 local indices, < 0 = a zero row"""
        o, L = self.ops, self._lib.lib()
        rows = rows.to(torch.int32).contiguous()
        out = torch.empty((rows.numel(), self.d), dtype=torch.float32, device=self.device)
        th, m, v, st = ((shard._P, shard._mP, shard._vP, shard.stP) if table == "P" else (shard._Q, shard._mQ, shard._vQ, shard.stQ))
        self._lib.check(L.macr_lazy_rows(rows.numel(), self.d, o._ptr(rows), o._ptr(th), o._ptr(m), o._ptr(v), o._ptr(st),
                                         o._ptr(shard.lazy_state), ctypes.byref(self.hyper), o._ptr(out), o._stream()))
        return out

    def lazy_flush(self, shard, tables="PQ"):
        o, L = self.ops, self._lib.lib()
        lz = self._lazy(shard)
        n_p = shard._P.shape[0] if "P" in tables else 0
        n_q = shard._Q.shape[0] if "Q" in tables else 0
        self._lib.check(L.macr_lazy_flush(self.d, n_p, n_q, o._ptr(shard._P), o._ptr(shard._Q),
                                          o._ptr(shard._mP), o._ptr(shard._vP), o._ptr(shard._mQ), o._ptr(shard._vQ),
                                          ctypes.byref(self.hyper), ctypes.byref(lz), o._stream()))

    def forward_and_bxb(self, shard, rows3, rank, world):
        o, L = self.ops, self._lib.lib()
        B = rows3.shape[1]
        self._lib.check(L.macr_shard_forward(self.kind, B, self.d, o._ptr(rows3), o._ptr(shard.w), o._ptr(shard.wu),
                                             o._ptr(self.ws), self.ws.numel(), o._stream()))
        if self.kind == self._lib.LOSS_NORMALBCE:
            return None                                     # no (B,B) term (macr_mf/model.py:277-287): nothing to sum over the ranks
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
        if not nbytes.value:
            return self.losses, None                        # normalbce: no branch vectors
        return self.losses, self._view(ptr, nbytes)         # losses; branch-vector partial rows (broadcast from rank 0)

    # ---- the split step: forward / backward of this rank's slice of the batch (macr_shard_*_slice)
    def route(self, shard, B, u, i, j, slice_ends):
        """(counts (W,W) int64 device, send_ref, recv_ref int64 (3B,)) of RowShardedMF.route by macr_shard_route: one launch"""
        W, dev = shard.world, u.device
        key = ("route_bufs", B)
        bufs = self.__dict__.get(key)
        if bufs is None:
            def bounds_of(own):
                if own.layout == "interleaved":
                    return None
                return torch.tensor([row_range(own.n_rows, r, W)[1] for r in range(W)], dtype=torch.int32, device=dev)
            bufs = self.__dict__[key] = dict(
                ends=torch.tensor(list(slice_ends), dtype=torch.int32, device=dev), bu=bounds_of(shard.own_u), bi=bounds_of(shard.own_i),
                counts=torch.zeros(W * W, dtype=torch.int32, device=dev), send=torch.zeros(3 * B, dtype=torch.int32, device=dev),
                recv=torch.zeros(3 * B, dtype=torch.int32, device=dev))
        b = bufs
        p = self.ops._ptr
        self._lib.check(self._lib.lib().macr_shard_route(
            B, W, shard.rank, p(u, torch.int32), p(i, torch.int32), p(j, torch.int32), p(b["bu"], allow_none=True),
            p(b["bi"], allow_none=True), p(b["ends"]), p(b["counts"]), p(b["send"]), p(b["recv"]), self.ops._stream(dev.index)))
        return b["counts"].view(W, W).long(), b["send"].long(), b["recv"].long()

    def slice_of(self, B, rank, world):
        t0, t1 = ctypes.c_int(), ctypes.c_int()
        self._lib.check(self._lib.lib().macr_shard_slice(B, self.d, rank, world, ctypes.byref(t0), ctypes.byref(t1)))
        return t0.value, t1.value

    def forward_slice(self, shard, B, t0, rows3_slice):
        self._reserve(B)
        o, L = self.ops, self._lib.lib()
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._lib.check(L.macr_shard_forward_slice(self.kind, B, self.d, t0, rows3_slice.shape[1], o._ptr(rows3_slice), o._ptr(shard.w),
                                                   o._ptr(shard.wu), ctypes.byref(ptr), ctypes.byref(nbytes), o._ptr(self.ws),
                                                   self.ws.numel(), o._stream()))
        return self._view(ptr, nbytes)                      # forward state + loss partials of the slice: sum over the ranks

    def bxb(self, B, rank, world):
        o, L = self.ops, self._lib.lib()
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._lib.check(L.macr_shard_bxb(B, self.d, rank, world, ctypes.byref(ptr), ctypes.byref(nbytes), o._ptr(self.ws),
                                         self.ws.numel(), o._stream()))
        return self._view(ptr, nbytes)

    def backward_slice(self, shard, B, t0, rows3_slice):
        o, L = self.ops, self._lib.lib()
        n = rows3_slice.shape[1]
        stage = torch.empty((3, n, self.d), dtype=torch.float32, device=self.device)
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._lib.check(L.macr_shard_backward_slice(self.kind, B, self.d, t0, n, o._ptr(rows3_slice), o._ptr(shard.w), o._ptr(shard.wu),
                                                    o._ptr(self.adam_pow), ctypes.byref(self.hyper), o._ptr(self.losses), o._ptr(stage),
                                                    ctypes.byref(ptr), ctypes.byref(nbytes), o._ptr(self.ws), self.ws.numel(),
                                                    o._stream()))
        return self.losses, stage, self._view(ptr, nbytes)

    def stage_rows(self, B):
        """(3*B, d) view of the staging buffer macr_shard_apply reads (row role*B + t)"""
        ptr = ctypes.c_void_p()
        self._lib.check(self._lib.lib().macr_shard_stage(B, self.d, ctypes.byref(ptr), self.ops._ptr(self.ws), self.ws.numel()))
        return self._view(ptr, ctypes.c_size_t(3 * B * self.d * 4)).view(3 * B, self.d)

    def apply(self, shard, u, i, j):
        o, L = self.ops, self._lib.lib()
        ou, oi = shard.own_u, shard.own_i
        head = (self.kind, u.numel(), self.d, ou.n, oi.n, ou.lo, ou.stride, oi.lo, oi.stride, o._ptr(u), o._ptr(i), o._ptr(j),
                o._ptr(shard._P), o._ptr(shard._Q), o._ptr(shard.w), o._ptr(shard.wu),
                o._ptr(shard._mP), o._ptr(shard._vP), o._ptr(shard._mQ), o._ptr(shard._vQ),
                o._ptr(shard.mw), o._ptr(shard.vw), o._ptr(shard.mwu), o._ptr(shard.vwu),
                o._ptr(shard.gP), o._ptr(shard.gQ), o._ptr(shard.tP), o._ptr(shard.tQ), ctypes.byref(self.hyper))
        tail = (o._ptr(self.ws), self.ws.numel(), o._stream())
        lz = self._lazy(shard)
        if lz is not None:
            self._lib.check(L.macr_shard_apply_lazy(*(head + (ctypes.byref(lz),) + tail)))
        else:
            self._lib.check(L.macr_shard_apply(*(head + tail)))


def _current_table(name):
    """property of RowShardedMF: a table or slot as the per-step dense pass would hold it (rows of THAT table the lazy pass left
    behind catch up first; the other table is left alone)"""
    def get(self):
        self.flush(name[-1])
        return getattr(self, name)
    return property(get)


class RowShardedMF(object):
    """This rank's shard of the MF model + its optimizer state.  P_full / Q_full (any rank-identical source) are only
    sliced at construction; afterwards a rank holds rows [u_lo,u_hi) of P and [i_lo,i_hi) of Q, nothing else."""

    def __init__(self, P_full, Q_full, w, wu, backend, rank=None, world=None, group=None, shards=None, layout="interleaved",
                 lazy_period=None):
        """shards=(P_shard, Q_shard, n_users, n_items): this rank's rows directly (P_full / Q_full are ignored) -- for
        tables whose full copy exists nowhere (10 M x 1 M rows, d = 128).  layout: see Owned.
        lazy_period: K of the lazy dense Adam pass (include/macr_hip.h: macr_lazy_adam) -- a step updates the batch's rows and
        one K-th of the shard, every row catching up its K steps in registers; 1 = the dense pass every step; None: the
        environment's MACR_LAZY_ADAM, else by shard size (lazy_period_for)."""
        r, ws = sharding.world()
        self.rank, self.world, self.group = (r if rank is None else rank), (ws if world is None else world), group
        self.n_users, self.n_items = (shards[2], shards[3]) if shards else (P_full.shape[0], Q_full.shape[0])
        self.own_u = Owned(self.n_users, self.rank, self.world, layout)
        self.own_i = Owned(self.n_items, self.rank, self.world, layout)
        clone = lambda t: t.clone().contiguous()
        if shards:
            self._P, self._Q = shards[0].contiguous(), shards[1].contiguous()
            assert self._P.shape[0] == self.own_u.n and self._Q.shape[0] == self.own_i.n
        else:
            self._P, self._Q = clone(self.own_u.take(P_full)), clone(self.own_i.take(Q_full))
        self.w, self.wu = clone(w.reshape(-1)), clone(wu.reshape(-1))
        self.collective_ms = None                   # bench: {"rows": [...], "partials": [...], "branch": [...]} event times
        self.split = os.environ.get("MACR_SHARD_SPLIT", "1") != "0"
        self.wire_rows = None
        z = torch.zeros_like
        self._mP, self._vP, self._mQ, self._vQ = z(self._P), z(self._P), z(self._Q), z(self._Q)
        self.mw, self.vw, self.mwu, self.vwu = z(self.w), z(self.w), z(self.wu), z(self.wu)
        self.gP, self.gQ = z(self._P), z(self._Q)
        dev = self._P.device
        self.tP = torch.zeros(self._P.shape[0], dtype=torch.int32, device=dev)
        self.tQ = torch.zeros(self._Q.shape[0], dtype=torch.int32, device=dev)
        self.backend = backend
        # lazy dense Adam: row stamps + the library's step counter / lr_t ring; `_stale` = some row is behind the current step
        if lazy_period is None:
            env = os.environ.get("MACR_LAZY_ADAM", "")
            lazy_period = int(env) if env else lazy_period_for(self._P.shape[0] + self._Q.shape[0], self._P.shape[1])
        if not getattr(backend, "lazy_capable", False):
            lazy_period = 1
        self.lazy_period, self._stale_tabs = max(1, int(lazy_period)), set()
        if self.lazy_period > 1:
            self.stP = torch.zeros(self._P.shape[0], dtype=torch.int32, device=dev)
            self.stQ = torch.zeros(self._Q.shape[0], dtype=torch.int32, device=dev)
            self.lazy_state = torch.zeros(1040, dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ the tables, as the per-step dense pass would hold them
    @property
    def _stale(self):
        return bool(self._stale_tabs)

    @_stale.setter
    def _stale(self, v):
        self._stale_tabs = {"P", "Q"} if v else set()

    def flush(self, tables="PQ"):
        """every row of the shard (of the named tables: "P", "Q", "PQ") brought to the current step; no-op unless the lazy pass
        left rows behind"""
        todo = [t for t in tables if t in self._stale_tabs]
        if todo:
            self.backend.lazy_flush(self, "".join(todo))
            self._stale_tabs -= set(todo)

    def rows(self, table, local_idx):
        """(n, d) rows local_idx (local indices) of table "P" / "Q" as of the current step WITHOUT bringing the table up to date:
        what an evaluation needs of the user table is its query users' rows, not a pass over every user"""
        if table in self._stale_tabs:
            return self.backend.lazy_rows(self, table, local_idx)
        # one meaning for an index < 0 in both branches: a zero row (lazy_rows' convention), never "the last row"
        idx = local_idx.long()
        out = getattr(self, "_" + table)[idx.clamp(min=0)]
        return out.masked_fill_((idx < 0).unsqueeze(1), 0.0)            # (no host synchronisation; `out` is a fresh gather)

    P, Q = _current_table("_P"), _current_table("_Q")
    mP, vP, mQ, vQ = _current_table("_mP"), _current_table("_vP"), _current_table("_mQ"), _current_table("_vQ")

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
            def via_host():
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            self._timed(name, via_host)
        else:
            self._timed(name, lambda: dist.all_reduce(t, group=self.group))

    def _broadcast(self, t, src=0, name="broadcast"):
        if self.world == 1 and not sharding.force_collectives():
            return
        if self._host_rig(t):
            def via_host():
                h = t.cpu()
                dist.broadcast(h, src, group=self.group)
                t.copy_(h)
            self._timed(name, via_host)
        else:
            self._timed(name, lambda: dist.broadcast(t, src, group=self.group))

    def _all_to_all(self, recv, send, recv_counts, send_counts, name):
        """rows (n, d): send_counts[q] rows go to rank q, recv_counts[q] arrive from it (python ints, the same on every rank)"""
        def run(r, s_):
            dist.all_to_all_single(r, s_, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=self.group)
        if self._host_rig(send):                    # test rig: several ranks on one GPU cannot use RCCL
            def via_host():
                r = torch.empty(recv.shape, dtype=recv.dtype)
                run(r, send.cpu())
                recv.copy_(r)
            self._timed(name, via_host)
        else:
            self._timed(name, lambda: run(recv, send))

    def route(self, u, i, j):
        """Who sends what to whom in the split step, from the batch alone (every rank computes the same tables).
        References are numbered role * B + t (role 0/1/2 = user / positive / negative row of position t).  Returns
          counts   (W, W) python ints: counts[q][p] = rows owner q sends to the rank whose slice holds their position, p
          send_ref (n_send,) references whose rows THIS rank owns, ordered by (destination, reference): what it sends
          recv_ref (n_recv,) references of THIS rank's slice ordered by (owner, reference): the order they arrive in
        The one host synchronisation of a split step is reading `counts` (W*W integers); callers that know the batch on the
        host (the CLI's sampler, the bench's batch pool) pass it to step_split instead."""
        B, W = u.numel(), self.world
        bounds = self._slice_bounds(B)
        rows = torch.cat([u, i, j]).long()
        # (one workgroup: it wins up to ~16 k triples -- 78-103 us against 296 at B = 8192 -- and loses beyond: 506-665 against 331 at 65 536)
        if (u.is_cuda and W <= 16 and B <= 16384 and hasattr(self.backend, "route")
                and os.environ.get("MACR_SHARD_ROUTE_TORCH", "0") != "1"):
            # ONE launch behind the C ABI (macr_shard_route) instead of the ~10 torch launches below: same tables, same order
            counts, send_ref, recv_ref = self.backend.route(self, B, u, i, j, [b[1] for b in bounds])
            return counts, send_ref, recv_ref, rows
        t1s = torch.tensor([b[1] for b in bounds], device=u.device)
        dest = torch.bucketize(torch.arange(B, device=u.device), t1s, right=True).repeat(3)             # (3B,)
        owner = torch.cat([self.own_u.owner_of(u.long()), self.own_i.owner_of(i.long()), self.own_i.owner_of(j.long())])
        counts = torch.bincount(owner * W + dest, minlength=W * W).view(W, W)
        mine = owner == self.rank
        send_ref = torch.argsort(torch.where(mine, dest, torch.full_like(dest, W)), stable=True)        # mine first, by destination
        in_slice = dest == self.rank
        recv_ref = torch.argsort(torch.where(in_slice, owner, torch.full_like(owner, W)), stable=True)  # my slice first, by owner
        return counts, send_ref, recv_ref, rows

    def route_counts_host(self, u, i, j):
        """the `counts` table of route() from a HOST copy of the batch (numpy int arrays): what a caller whose sampler runs on
        the host hands to step() so that the split step never waits for the device"""
        import numpy as np
        B, W = len(u), self.world
        t1s = np.asarray([b[1] for b in self._slice_bounds(B)])
        dest = np.tile(np.searchsorted(t1s, np.arange(B), side="right"), 3)
        ids = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64))
        owner = torch.cat([self.own_u.owner_of(ids(u)), self.own_i.owner_of(ids(i)), self.own_i.owner_of(ids(j))]).numpy()
        return np.bincount(owner * W + dest, minlength=W * W).reshape(W, W).tolist()

    def _slice_bounds(self, B):
        key = ("slices", B)
        if key not in self.__dict__:
            self.__dict__[key] = [self.backend.slice_of(B, r, self.world) for r in range(self.world)]
        return self.__dict__[key]

    def step_split(self, u, i, j, counts=None):
        """One step with forward and backward SPLIT over the ranks (branch losses): rank p handles the positions of its
        (B,B) row blocks.  Per rank and step: all-to-all of the 3B/W rows its slice needs (from their owners), an all-reduce of
        the forward scalars (7 floats per position), the all-reduce of the (B,B) partial sums, an all-reduce of the
        branch-vector gradient partials, all-to-all of the 3B/W gradient rows back to the owners, then the local Adam pass.
        counts: the (W, W) table of route() as python ints when the caller knows the batch on the host (no synchronisation)."""
        be, W, B = self.backend, self.world, u.numel()
        cnt, send_ref, recv_ref, rows = self.route(u, i, j)
        counts = cnt.tolist() if counts is None else counts         # (the step's one host synchronisation when not given)
        t0, t1 = self._slice_bounds(B)[self.rank]
        n = t1 - t0
        send_counts = [counts[self.rank][p] for p in range(W)]       # rows I own, by the rank whose slice needs them
        recv_counts = [counts[q][self.rank] for q in range(W)]       # rows of my slice, by owner
        n_send, n_recv = sum(send_counts), sum(recv_counts)
        assert n_recv == 3 * n
        send_ref, recv_ref = send_ref[:n_send], recv_ref[:n_recv]
        d = self._P.shape[1]
        # 1. rows of my slice from their owners
        sr = rows[send_ref]
        if self.lazy_period > 1:                        # rows as of this step, brought up to date on their way out (nothing written)
            user = send_ref < B
            neg = torch.full_like(sr, -1)
            send = (be.lazy_rows(self, "P", torch.where(user, self.own_u.local_index(sr), neg)) +
                    be.lazy_rows(self, "Q", torch.where(user, neg, self.own_i.local_index(sr))))     # (x + 0: exact)
        else:
            is_user = (send_ref < B).unsqueeze(1)
            from_p = self._P[self.own_u.local_index(sr).clamp(0, max(self._P.shape[0] - 1, 0))]     # (both tables are indexed for every
            from_q = self._Q[self.own_i.local_index(sr).clamp(0, max(self._Q.shape[0] - 1, 0))]     # reference: n_send rows each, no branch)
            send = torch.where(is_user, from_p, from_q)
        recv = torch.empty((n_recv, d), dtype=self._P.dtype, device=self._P.device)
        self._all_to_all(recv, send.contiguous(), recv_counts, send_counts, "rows_a2a")
        # arrival order -> (role, position in the slice)
        role, t = recv_ref // B, recv_ref % B
        slot = role * n + (t - t0)
        rows3 = torch.empty((3 * n, d), dtype=self._P.dtype, device=self._P.device)
        rows3[slot] = recv
        rows3 = rows3.view(3, n, d)
        # 2. forward of the slice; its scalars and loss partials summed into the whole batch's
        self._all_reduce(be.forward_slice(self, B, t0, rows3), "fwd")
        # 3. my (B,B) row blocks; partial row / column sums over the ranks
        self._all_reduce(be.bxb(B, self.rank, W), "partials")
        # 4. backward of the slice; branch-vector gradient partials summed (identical on every rank afterwards)
        losses, stage, branch = be.backward_slice(self, B, t0, rows3)
        self._all_reduce(branch, "branch")
        # 5. gradient rows back to the owners, into the staging rows macr_shard_apply reads
        back = torch.empty((n_send, d), dtype=self._P.dtype, device=self._P.device)
        self._all_to_all(back, stage.view(3 * n, d)[slot].contiguous(), send_counts, recv_counts, "grads_a2a")
        be.stage_rows(B)[send_ref] = back
        be.apply(self, u, i, j)
        self._stale = self.lazy_period > 1
        self.wire_rows = (n_send - send_counts[self.rank]) + (n_recv - recv_counts[self.rank])   # rows that crossed ranks, one way each
        return losses

    # ------------------------------------------------------------------ one step
    def step(self, u, i, j, counts=None):
        """u, i, j: the SAME batch on every rank (int32, global row ids).  Returns {loss, mf_loss, reg_loss} (3,).
        With several ranks and a branch loss the step is the SPLIT one (step_split; MACR_SHARD_SPLIT=0: the replicated
        forward / backward around one all-reduce of the batch's rows, below); `counts`: see step_split."""
        be = self.backend
        if (self.world > 1 and self.split and hasattr(be, "slice_of") and getattr(be, "kind", 1) != 0):   # (kind 0 = normalbce: no (B,B) term)
            return self.step_split(u, i, j, counts)
        rows3 = be.gather(self, u, i, j)
        self._all_reduce(rows3, "rows")                           # 1. the batch's rows, everywhere
        partials = be.forward_and_bxb(self, rows3, self.rank, self.world)
        if partials is not None:
            self._all_reduce(partials, "partials")                # 3. row / column sums of the (B,B) term
        losses, branch = be.backward(self, rows3)
        if branch is not None:
            self._broadcast(branch, 0, "branch")                  # 4. one copy of the branch-vector gradients
        be.apply(self, u, i, j)                                   # 5. local segment reduce + dense Adam on the shard
        self._stale = self.lazy_period > 1
        return losses

    def full_tables(self):
        """(P, Q) reassembled on every rank (tests / checkpoints; not on the training path)."""
        def cat(local, own):
            if self.world == 1:
                return local.clone()
            full = torch.zeros((own.n_rows, local.shape[1]), dtype=local.dtype, device=local.device)
            full[own.global_ids(local.device)] = local
            self._all_reduce(full)
            return full
        return cat(self.P, self.own_u), cat(self.Q, self.own_i)
