"""Full-catalogue top-K evaluator on the HIP path.

Replaces the per-batch  sess.run(ratings) -> host  -> per-user Python/C++ ranking
of the reference:
  * macr_mf/train.py  test() :162-311  (+ test_one_user :119-138, metrics :32-117)
  * macr_lightgcn/utility/batch_test.py  test() :26-162  (+ evaluator/cpp)
with: branch sigmoids -> fused scoring/mask/top-K (MFMA) -> merge of splits ->
[multi-GPU: one RCCL all-gather of the per-shard top-K + merge] -> metrics
kernel -> column means.  Nothing U x N ever leaves the device (or exists).

Both reference evaluators batch the users by BATCH_SIZE; every per-user result
is independent of the batching, so all query users are ranked in one pass.
"""
import os

import numpy as np
import torch

from . import ops, sharding
from . import _lib as _lib_consts

_LP_FILTERS = ("bf16", "f16")       # reduced-precision candidate filters (fp32 re-scoring; include/macr_hip.h MACR_EVAL_FILTER_*)


class Evaluator(object):
    def __init__(self, mask_lists, gt_lists, n_items, device):
        """mask_lists[q]: the train items of query user q (excluded from ranking, train.py:132-133 /
        batch_test.py:124-129); gt_lists[q]: its test (or valid) items."""
        assert len(mask_lists) == len(gt_lists)
        self.n_queries = len(gt_lists)
        self.n_items = n_items
        device = torch.device(device)
        self.device = device
        self.mask = ops.CSR.from_lists(mask_lists, device)
        self.max_queries_per_pass = 131072       # bounds the ranking workspace (~1.5 GB of candidate lists)
        self.use_graph = os.environ.get("MACR_EVAL_GRAPH", "1") != "0"    # replay the evaluation as one HIP graph (_means)
        self.use_seeds = os.environ.get("MACR_EVAL_SEEDS", "1") != "0"    # thresholds from the previous top K (rank_local)
        # candidate filter of the listing pass, an argument of every ranking call (include/macr_hip.h MACR_EVAL_FILTER_*): "f16" (default, round 6) =
        # one fp16 number per operand on the fp16 matrix cores, "bf16" = two-term bf16 products on the bf16 matrix cores -- both
        # with fp32 re-scoring of the best candidates -- "f32" = fp32 products throughout.  The ranking is the fp32 ranking bit
        # for bit whichever it is; MACR_EVAL_FILTER in the environment overrides the default; the policy below steps down
        # f16 -> bf16 -> f32 where a model's scores are packed closer than a filter resolves.
        self.filter = os.environ.get("MACR_EVAL_FILTER", "f16").strip().lower()
        ops.eval_filter_code(self.filter)         # a typo in the environment is refused here, by name
        # One GPU, graph replays: an evaluation launches the FIRST ROUND of the ranking only and writes its means and the
        # ranking's stats straight into pinned host memory; the host, which waits for the means anyway, sees whether a
        # candidate list overflowed or the seeds were stale and, in that rare case, replays the REST of the ranking (repair
        # round, fallback: macr_score_topk_repair_round) on the same workspace and outputs.  Saves the launches that find
        # nothing to do (~25 us of a 0.45 ms evaluation on the Gowalla shape) and both result copies.
        # MACR_EVAL_OPTIMISTIC=0: the complete sequence always.
        self.optimistic = os.environ.get("MACR_EVAL_OPTIMISTIC", "1") != "0"
        # Launch folds.  "p" (default): the branch factors and the ranking's workspace initialisation are one launch
        # (macr_score_topk_prologue): -6 .. -8 us per graph-replayed evaluation on the Gowalla shape, same box.  "m": the
        # MF metrics and their means as one launch (macr_metrics_mf_mean): 20 us against 12.6 + 7.5 us for the two kernels it
        # replaces -- no gain under graph replay (profiles/r05_eval_fold_ab.txt), so off unless asked for.  "1": both, "0": none.
        fold = os.environ.get("MACR_EVAL_FOLD", "p").strip().lower()
        self.fold_prologue, self.fold_metrics = fold in ("1", "p"), fold in ("1", "m")
        self.fold_prep = os.environ.get("MACR_EVAL_FOLD_PREP", "1") != "0"       # (A/B switch of macr_score_topk_prologue_prep)
        self._topk_mode = None                    # None: the complete call; "first" / "repair": its two halves
        self._repair_bufs = None
        self._last_entry = None
        self._host_out = {}
        self.fast_stats = {"fast": 0, "redone": 0}
        self._graphs = {}
        self._graph_misses = 0
        # seeding policy: thresholds come from the previous ranking unless that went badly last time
        self._seeded_now = True                   # what the ranking about to be launched does (if it has seeds at all)
        self._seed_skip, self._seed_backoff = 0, 1
        # filter policy (the same shape): a bf16-filter evaluation that ended in the exact kernel -- candidate lists overflowed in
        # both rounds: scores packed tighter at the top than the filter's error bound resolves, e.g. a catalogue of a million
        # barely trained items under c = 40 -- cost 2-3x an fp32-filter evaluation; the next 1, 2, 4 ... 16 evaluations take the
        # fp32 filter before bf16 is tried again.  The ranking is the same either way.
        self._bf16_skip, self._bf16_backoff = 0, 1
        # ... and one tier above it: the fp16 filter's margin is 12x the bf16 filter's.  An UNSEEDED fp16 evaluation that had
        # to list query blocks again (lists that overflowed under a threshold less that margin: scores at the top closer
        # together than fp16 resolves, e.g. (y - c) sig_i sig_u with c = 30 on barely trained rows of d = 128), or one that
        # ended in the exact kernel, sends the next 1, 2, 4 ... 16 evaluations to the bf16 filter (whose own back-off leads on to fp32).
        self._f16_skip, self._f16_backoff = 0, 1
        self._stats = torch.zeros(2, dtype=torch.int32, device=device)          # macr_score_topk stats of the last ranking
        self._stats_host = torch.zeros(2, dtype=torch.int32)
        self._stats_first = torch.zeros(2, dtype=torch.int32)          # written by the first-round ranking's own kernel
        if device.type == "cuda":
            self._stats_host = self._stats_host.pin_memory()
            self._stats_first = self._stats_first.pin_memory()
        self._stats_evt = None
        self._last_seeded = False
        self.gt = ops.CSR.from_lists(gt_lists, device)
        # (lo, hi): the `items_tab` the methods below receive is ALREADY this rank's shard, rows [lo, hi) of a catalogue
        # whose full table exists nowhere (row-sharded training, BASELINE configs[4]); None: a replicated full table
        self.local_items_range = None
        # the same for a STRIDED shard (set_local_items): local row l is item lo + l * stride
        self._local_own = None
        self._mask_lists = mask_lists

    def set_local_items(self, own):
        """`items_tab` is this rank's shard of an item table sharded like the training state (sharded_train.Owned: local
        row l = item own.lo + l * own.stride).  A contiguous shard is an item offset; an interleaved one is ranked in LOCAL
        ids against the owned part of every query's train list and its ids are mapped back before the shards meet."""
        if own.stride == 1:
            self.local_items_range = (own.lo, own.lo + own.n)
            self._local_own = None
            return
        self.local_items_range = None
        self._local_own = own
        local_lists = [[(g - own.lo) // own.stride for g in row if g >= own.lo and (g - own.lo) % own.stride == 0
                        and (g - own.lo) // own.stride < own.n] for row in self._mask_lists]
        self._mask_local = ops.CSR.from_lists(local_lists, self.device)
        self.__dict__.pop("_seeds", None)

    def _shard(self, items_tab):
        """(lo, hi, this rank's rows of the item table)"""
        if self._local_own is not None:
            assert items_tab.shape[0] == self._local_own.n
            return 0, self._local_own.n, items_tab
        if self.local_items_range is not None:
            lo, hi = self.local_items_range
            assert items_tab.shape[0] == hi - lo
            return lo, hi, items_tab
        rank, ws = sharding.world()
        lo, hi = sharding.item_shard_range(items_tab.shape[0], rank, ws)
        return lo, hi, items_tab[lo:hi]

    # ------------------------------------------------------------------ ranking
    def rank_local(self, kind, users_tab, user_ids, items_tab, K, w=None, wu=None, c=0.0):
        """This rank's item shard: (val, idx) of shape (U,K) with GLOBAL item ids.
        users_tab/items_tab: full embedding tables (replicated on every rank); a rank scores only its contiguous
        item shard."""
        ws = sharding.world()[1]
        lo, hi, items_local = self._shard(items_tab)
        sig_u = sig_i = None
        U = self.n_queries
        both = kind in (ops.SCORE_RUBI_BOTH, ops.SCORE_DIRECT_MINUS_BOTH)
        # one ranking call that initialises its own workspace: the branch factors and that initialisation are ONE launch
        # (macr_score_topk_prologue, below, once it is known whether the call is seeded)
        fold = (kind != ops.SCORE_NORMAL and U <= self.max_queries_per_pass and self._topk_mode != "repair"
                and items_local.shape[1] == users_tab.shape[1] and self.fold_prologue)
        if fold:
            pass
        elif both and items_local.shape[1] == users_tab.shape[1]:
            # sigmoid(e_i . w), sigmoid(e_u . w_user) (model.py:141-142,:199-201) in one launch
            sig_i, sig_u = ops.branch_sigmoid2(items_local, w, None, users_tab, wu, user_ids)
        elif kind != ops.SCORE_NORMAL:
            sig_i = ops.branch_sigmoid(items_local, w)              # sigmoid(e_i . w)      model.py:141-142,:199-201
            if both:
                sig_u = ops.branch_sigmoid(users_tab, wu, user_ids)     # sigmoid(e_u . w_user) model.py:199,:201
        if U <= self.max_queries_per_pass:
            # Seeds: the ids this shard returned last time (same queries, tables that moved by a few training steps).
            # Their exact current scores bound every query's K-th best score from below far more tightly than a
            # sampling pass does, for a tenth of its time (k_tau_seed); the ranking itself does not depend on them.
            # Seeds the tables have moved away from (early epochs) cost a repair round, so the evaluator watches how
            # many query blocks were listed twice (_seed_feedback) and goes back to the sampling pass for a while.
            seeds = self.__dict__.setdefault("_seeds", {})
            use = (self.use_seeds and ws == 1 and K <= _lib_consts.MAX_TOPK_FUSED       # (the wide ranking takes no seeds)
                   and self._shape_uses_seeds(hi - lo, items_tab.shape[1]))
            seed = seeds.get((K, lo, hi)) if use else None
            seeded = self._ranked_seeded = seed is not None and self._seeded_now
            if use and seed is None:
                seed = seeds[(K, lo, hi)] = torch.full((U, ops.SEED_WIDTH), -1, dtype=torch.int32, device=self.device)
            # the ranking leaves its best SEED_WIDTH candidates per query in `seed` (in place): the next ranking's seeds
            mask = self._mask_local if self._local_own is not None else self.mask
            mode = self._topk_mode
            # under the fp16 filter the prologue also writes the listing pass's operand copies: every row read once (the tables are
            # cold after a log interval of training) -- not for a catalogue small enough to list everything, which runs no filter
            fold_prep = (fold and self.fold_prep and self.filter_now == "f16" and K <= _lib_consts.MAX_TOPK_FUSED
                         and self._shape_uses_seeds(hi - lo, items_tab.shape[1]))
            if fold_prep:
                sig_i, sig_u = ops.score_topk_prologue_prep(kind, users_tab, user_ids, items_local, K, w, wu if both else None, c,
                                                            seeded_first_round=seeded and mode == "first")
            elif fold:
                sig_i, sig_u = ops.score_topk_prologue(users_tab, user_ids, items_local, K, w, wu if both else None,
                                                       seeded_first_round=seeded and mode == "first", filter=self.filter_now)
            vals, idx = ops.score_topk(kind, users_tab, user_ids, items_local, K, sig_u, sig_i, c, mask, lo,
                                       seed=seed if seeded else None, seed_out=seed,
                                       stats=self._stats_first if mode else self._stats, first_round=mode == "first",
                                       repair_of=self._repair_bufs if mode == "repair" else None, filter=self.filter_now,
                                       ws_ready=fold, prep_ready=fold_prep)
        else:
            # the ranking workspace (candidate lists, mask bitmap) grows with the number of queries: rank them in
            # chunks; every query is independent of the chunking
            parts = []
            for a in range(0, U, self.max_queries_per_pass):
                b = min(U, a + self.max_queries_per_pass)
                uid = user_ids[a:b] if user_ids is not None else torch.arange(a, b, dtype=torch.int32, device=self.device)
                mask = self._mask_local if self._local_own is not None else self.mask
                parts.append(ops.score_topk(kind, users_tab, uid, items_local, K, None if sig_u is None else sig_u[a:b],
                                            sig_i, c, mask.row_range(a, b), lo, filter=self.filter_now))
            vals = torch.cat([p[0] for p in parts], dim=1)
            idx = torch.cat([p[1] for p in parts], dim=1)
        if self._local_own is not None:           # local row -> item id (the order by id within the shard is the same either way)
            own = self._local_own
            idx = torch.where(idx >= 0, idx * own.stride + own.lo, idx)
        return vals, idx

    def _seed_feedback(self):
        """Decide, before a ranking is launched, whether it takes its thresholds from the seeds.  The previous seeded
        ranking reports whether blocks of 256 queries had to be listed twice because a seeded threshold was too loose
        (macr_score_topk stats).  The repair round costs about as much as an unseeded ranking however few blocks it
        lists, so one of them means the model still moves too far between two evaluations for seeds to pay (early
        epochs): the next 1, 2, 4 ... 16 evaluations use the sampling pass before seeds are tried again.  The choice only
        moves time around: every mode returns the same ranking."""
        # Several ranks: no seeds.  The policy's state is per shard, so ranks would switch between the seeded and the
        # sampled launch sequence -- and capture the other graph, with its extra warm-up collectives -- at different
        # evaluations: mismatched all-gathers.  (A 1/8 shard's sampling pass is 13 us; there is little to win.)
        if not self.use_seeds or sharding.world()[1] > 1:
            self._seeded_now = False
            return False
        if self._stats_evt is not None:
            self._stats_evt.synchronize()         # normally long complete: the caller has read the previous metrics
            self._stats_evt = None
            if self._last_seeded:
                relisted, fell_back = int(self._stats_host[0]), int(self._stats_host[1])
                if relisted > self._relist_tolerance():
                    self._seed_skip = self._seed_backoff
                    self._seed_backoff = min(16, 2 * self._seed_backoff)
                else:
                    self._seed_backoff = 1
        if self._seed_skip > 0:
            self._seed_skip -= 1
            self._seeded_now = False
        else:
            self._seeded_now = True
        return self._seeded_now

    @property
    def filter_now(self):
        """the candidate filter of the ranking about to be launched: `filter`, or "f32" while a reduced-precision filter
        ("bf16", "f16") is backed off"""
        if self.filter == "f16" and self._f16_skip == 0:
            return "f16"
        if self.filter in _LP_FILTERS:
            return "f32" if self._bf16_skip > 0 else "bf16"
        return self.filter

    def _relist_tolerance(self):
        """blocks of 256 queries a seeded ranking may list twice before the seeds count as stale: none up to 63 blocks (a repair
        round costs what an unseeded ranking costs there), one per 64 blocks beyond -- on 100 000 queries a handful of
        re-listed blocks is a few queries with degenerate scores (every item tied), not a model that moved away from its seeds,
        and the sampling pass would cost every block more than their repair does"""
        return ((self.n_queries + 255) // 256) // 64

    def _shape_uses_seeds(self, n_local, d):
        """False for shards small enough that the ranking lists every unmasked item (no thresholds to seed)"""
        key = (n_local, d)
        cache = self.__dict__.setdefault("_uses_seeds", {})
        if key not in cache:
            from . import _lib
            cache[key] = bool(_lib.lib().macr_score_topk_uses_seeds(self.n_queries, n_local, d))
        return cache[key]

    def _has_seeds(self, K, n_items):
        if self._local_own is not None:
            return (K, 0, self._local_own.n) in self.__dict__.get("_seeds", {})
        if self.local_items_range is not None:
            return (K,) + tuple(self.local_items_range) in self.__dict__.get("_seeds", {})
        rank, ws = sharding.world()
        return (K,) + tuple(sharding.item_shard_range(n_items, rank, ws)) in self.__dict__.get("_seeds", {})

    def _stats_readback(self, seeded):
        """after a ranking was launched: its stats travel to the host behind it (no synchronisation here)"""
        self._last_seeded = seeded
        if seeded and self.device.type == "cuda":           # only a seeded ranking's stats steer anything
            self._stats_host.copy_(self._stats, non_blocking=True)
            self._stats_evt = torch.cuda.Event()
            self._stats_evt.record()

    def rank(self, kind, users_tab, user_ids, items_tab, K, w=None, wu=None, c=0.0, fill_masked=False):
        """Top-K item ids for every query user: (val (U,K), idx (U,K), cnt (U,)); the shards' top-K are all-gathered
        (one collective) and merged."""
        self._seed_feedback()
        self._ranked_seeded = False
        vals, idx = self.rank_local(kind, users_tab, user_ids, items_tab, K, w, wu, c)
        self._stats_readback(self._ranked_seeded)
        fill = self.mask if fill_masked else None
        if sharding.world()[1] == 1:
            return ops.topk_merge(vals, idx, fill)
        lv, li, _ = ops.topk_merge(vals, idx)                        # merge this shard's splits
        gv, gi = sharding.gather_topk(lv, li)                        # (W,U,K) over RCCL / xGMI
        return ops.topk_merge(gv, gi, fill)

    # ------------------------------------------------------------------ MF flavour
    def test_mf(self, kind, users_tab, user_ids, items_tab, Ks, w=None, wu=None, c=0.0):
        """-> {'precision','recall','ndcg','hit_ratio'}: np.ndarray(len(Ks)) float64, the mean over the
        query users (train.py:286-290 accumulates re[...]/n_test_users)."""
        m = self._means("mf", kind, users_tab, user_ids, items_tab, tuple(Ks), w, wu, c).cpu().numpy()
        return {'precision': m[0].copy(), 'recall': m[1].copy(), 'ndcg': m[2].copy(), 'hit_ratio': m[3].copy()}

    def _finish(self, flavour, vals, idx, Ks, out=None):
        """(W,U,K) lists (splits of one shard, or the gathered shards) -> column means of the per-user metrics.
        out: optional pinned host tensor the last kernel writes the means to."""
        if flavour == "mf":
            # metrics and their means over the queries in one launch (macr_metrics_mf_mean); its scratch is this evaluator's
            # own (a captured graph bakes the address in)
            if not self.fold_metrics:
                if vals.shape[0] == 1:
                    return ops.colmean(ops.metrics_mf(idx[0], None, self.gt, list(Ks)), out=out)
                _, ix, cnt = ops.topk_merge(vals, idx)
                return ops.colmean(ops.metrics_mf(ix, cnt, self.gt, list(Ks)), out=out)         # (U,4,nK) float64 -> (4,nK)
            mws = self.__dict__.setdefault("_mean_ws", {})
            if len(Ks) not in mws:
                mws[len(Ks)] = ops.metrics_mf_mean_workspace(idx.shape[1], len(Ks), idx.device)
            if vals.shape[0] == 1:
                # one sorted list per query: nothing to merge (the metrics kernel counts a list's ids itself)
                return ops.metrics_mf_mean(idx[0], None, self.gt, list(Ks), mws[len(Ks)], out=out)
            _, ix, cnt = ops.topk_merge(vals, idx)
            return ops.metrics_mf_mean(ix, cnt, self.gt, list(Ks), mws[len(Ks)], out=out)   # (4,nK) float64
        if vals.shape[0] == 1 and vals.shape[2] <= _lib_consts.MAX_TOPK and self._local_own is None:
            # one sorted list per query: the metrics kernel completes short lists with the masked ids itself (-inf fill,
            # batch_test.py:124-134) -- no merge launch
            return ops.colmean(ops.metrics_foldout(idx[0], self.gt, hr_in_ap_slot=True, fill_mask=self.mask), out=out)
        _, ix, _ = ops.topk_merge(vals, idx, self.mask)                                      # -inf fill, batch_test.py:124-134
        return ops.colmean(ops.metrics_foldout(ix, self.gt, hr_in_ap_slot=True), out=out)   # (U,5*max_top) fp32

    def _direct(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c):
        vals, idx = self.rank_local(kind, users_tab, user_ids, items_tab, max(Ks), w, wu, c)
        if sharding.world()[1] > 1:
            lv, li, _ = ops.topk_merge(vals, idx)
            vals, idx = sharding.gather_topk(lv, li)
        return self._finish(flavour, vals, idx, Ks)

    def _c_scalar(self, c):
        """The evaluator's device copy of c: kernels read it at run time (macr_score_topk c_dev), so the captured
        graph of an evaluation serves every c of a sweep -- only this scalar is rewritten between replays."""
        if getattr(self, "_c_dev", None) is None:
            self._c_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._c_host = None
        if self._c_host != float(c):
            self._c_dev.fill_(float(c))
            self._c_host = float(c)
        return self._c_dev

    def _means(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c):
        """The device part of an evaluation.  An evaluator ranks the same queries against the same (in-place updated)
        tables every epoch, and the ~14 launches of one evaluation are issued from Python between two host
        synchronisations: on one GPU the sequence is captured once into a HIP graph and replayed (one launch instead
        of ~90 us of launch gaps per evaluation).  c is not part of the sequence: the kernels read it from a device
        scalar at run time, so the c sweep of the tuners (tune.py:545-578) replays ONE graph.  Anything else that changes
        the sequence -- other tensors, K -- is another graph; `use_graph = False` launches directly.  With several ranks
        the sequence is two graphs around the one collective (all-gather of the shards' top-K)."""
        c = self._c_scalar(c)
        world = sharding.world()[1]
        if (self.optimistic and self.use_graph and world == 1 and self.device.type == "cuda"
                and self.n_queries <= self.max_queries_per_pass and max(Ks) <= _lib_consts.MAX_TOPK_FUSED):
            return self._means_optimistic(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c)
        self._last_info = None
        # (no seeds yet for this K and shard: the first ranking samples, and leaves them)
        seeded = self._seed_feedback() and self.n_queries <= self.max_queries_per_pass and self._has_seeds(max(Ks), items_tab.shape[0])
        self._seeded_now = seeded        # the warm-up run of a capture creates the seeds: the capture itself must not pick them up
        try:
            return self._means_launch(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c, world, seeded)
        finally:
            self._stats_readback(seeded)

    def _means_optimistic(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c):
        """First round only, results in pinned host memory; the rest of the ranking when the first round says so."""
        if self._stats_evt is not None:           # (a stats copy of the complete path still in flight: not needed any more)
            self._stats_evt = None
        seeded = (self.use_seeds and self._seed_skip == 0 and self._has_seeds(max(Ks), items_tab.shape[0]))
        if self.use_seeds and self._seed_skip > 0:
            self._seed_skip -= 1
        used = self.filter_now
        try:
            return self._means_optimistic_run(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c, seeded)
        finally:
            info = getattr(self, "_last_info", None) or {}
            if used == "f16":
                if info.get("exact_fallback") or (not seeded and info.get("query_blocks_relisted")):
                    self._f16_skip = self._f16_backoff
                    self._f16_backoff = min(16, 2 * self._f16_backoff)
                elif not info.get("redone"):
                    self._f16_backoff = 1
            else:
                if self.filter == "f16" and self._f16_skip > 0 and not (used == "f32" and self._bf16_skip > 0):
                    self._f16_skip -= 1                  # (an evaluation spent in the bf16 tier; fp32 ones count for the bf16 tier's own wait)
                if used == "bf16":
                    if info.get("exact_fallback"):
                        self._bf16_skip = self._bf16_backoff
                        self._bf16_backoff = min(16, 2 * self._bf16_backoff)
                    elif not info.get("redone"):
                        self._bf16_backoff = 1
                elif self._bf16_skip > 0:
                    self._bf16_skip -= 1
            if getattr(self, "_last_info", None) is not None:
                self._last_info["filter"] = used

    def _means_optimistic_run(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c, seeded):
        self._seeded_now = seeded
        self._topk_mode = "first"
        try:
            out = self._means_launch(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c, 1, seeded, mode="first")
            first_entry = self._last_entry
        finally:
            self._topk_mode = None
        torch.cuda.current_stream().synchronize()
        self._last_seeded = False                 # (nothing for _seed_feedback to read later)
        relisted = int(self._stats_first[0])
        self._last_info = {"seeded": bool(seeded), "query_blocks_relisted": relisted, "exact_fallback": 0, "redone": relisted != 0}
        if self.__dict__.pop("_complete_ran", False):
            # graph replay was just switched off and the COMPLETE sequence ran in place of the first round: what it relisted and
            # whether it fell back are in its own statistics (the bf16 back-off must see a fallback there too)
            st = self._stats.tolist()
            self._last_info.update(query_blocks_relisted=st[0], exact_fallback=st[1])
            return out.clone()
        if relisted == 0:
            self.fast_stats["fast"] += 1
            if seeded:
                self._seed_backoff = 1
            return out.clone()
        # a list overflowed or seeds were stale: the repair round (and, behind it, the exact fallback) on the first round's
        # workspace and outputs -- what the complete call would have launched
        self.fast_stats["redone"] += 1
        if seeded and relisted > self._relist_tolerance():
            self._seed_skip = self._seed_backoff
            self._seed_backoff = min(16, 2 * self._seed_backoff)
        if first_entry is None or len(first_entry) < 4:      # (no graph of the first round: it ran as the complete call already)
            return out.clone()
        self._topk_mode, self._repair_bufs = "repair", first_entry[3]
        try:
            out = self._means_launch(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c, 1, seeded, mode="repair")
        finally:
            self._topk_mode, self._repair_bufs = None, None
        torch.cuda.current_stream().synchronize()
        self._last_info["exact_fallback"] = int(self._stats_first[1])
        return out.clone()

    def last_eval_info(self):
        """What the last evaluation did: {"seeded": its thresholds came from the previous ranking's candidates,
        "query_blocks_relisted": blocks of 256 queries whose lists overflowed / whose seeds were stale, "exact_fallback",
        "redone": the first round did not stand and the complete sequence ran (optimistic mode)}.  Synchronises."""
        if getattr(self, "_last_info", None) is not None:
            return dict(self._last_info)
        st = self._stats.tolist()
        return {"seeded": bool(self._last_seeded), "query_blocks_relisted": st[0], "exact_fallback": st[1], "redone": False}

    def _means_launch(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c, world, seeded, mode=None):
        self._last_entry = None
        if not self.use_graph:
            return self._direct(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c)
        host_out = None
        if mode:
            hk = (flavour, Ks)
            if hk not in self._host_out:
                shape = (4, len(Ks)) if flavour == "mf" else (5 * max(Ks),)
                self._host_out[hk] = torch.zeros(shape, dtype=torch.float64).pin_memory()
            host_out = self._host_out[hk]
        key = (flavour, mode, self.filter_now, kind, seeded, users_tab.data_ptr(), None if user_ids is None else user_ids.data_ptr(), items_tab.data_ptr(),
               Ks, None if w is None else w.data_ptr(), None if wu is None else wu.data_ptr(),
               torch.cuda.current_stream().cuda_stream, world)
        entry = self._graphs.get(key)
        if entry is None:
            # capturing costs about two evaluations: callers that keep changing the sequence are better off launching directly
            # Misses are counted per evaluation SHAPE (tables, query set, score kind, Ks, stream): one shape legitimately needs up
            # to 8 graphs (first / repair round x candidate filter x seeded or not), so a run that evaluates two query sets and
            # meets a back-off must not lose graph replay.  A caller whose tensors change from call to call shows up as many
            # shapes, or as one shape asking for more graphs than its key space has.
            shape_key = (flavour, kind) + key[5:]
            misses = self.__dict__.setdefault("_graph_misses_by_shape", {})
            misses[shape_key] = misses.get(shape_key, 0) + 1
            self._graph_misses += 1
            if misses[shape_key] > 8 or len(misses) > 8:
                self.use_graph = False
                self._graphs.clear()
                if mode == "first":
                    self._topk_mode = None        # (the complete sequence: its result needs no check)
                    self._stats_first.zero_()
                    self._complete_ran = True     # (its own statistics are in self._stats: _means_optimistic_run reads them)
                return self._direct(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c)
            self._direct(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c)     # warm-up: allocations, caches, attributes
            torch.cuda.synchronize()
            K = max(Ks)
            if world == 1:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    vals, idx = self.rank_local(kind, users_tab, user_ids, items_tab, K, w, wu, c)
                    out = self._finish(flavour, vals, idx, Ks, out=host_out)
                stages = (g, None, None, None)
                # what a repair round must continue on: the first round's outputs AND the workspace it was captured with
                # (the per-device cache is regrown whenever a larger evaluator asks: ops._topk_workspace)
                first_bufs = (vals, idx, ops._topk_ws_cache.get(items_tab.device))
            else:
                # several ranks: the collective stays outside -- one graph up to this shard's merged lists, the
                # all-gather (RCCL), one graph from the gathered lists to the means
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    vals, idx = self.rank_local(kind, users_tab, user_ids, items_tab, K, w, wu, c)
                    lv, li, _ = ops.topk_merge(vals, idx)
                gv, gi = sharding.gather_topk(lv, li)                    # static inputs of the second graph
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb):
                    out = self._finish(flavour, gv, gi, Ks)
                stages = (ga, (lv, li), (gv, gi), gb)
                first_bufs = None
            # the graphs bake in the addresses of everything they touched: keep the inputs and the cached scratch
            # (ranking workspace, mask bitmaps) alive for as long as they exist, whatever the caches do later
            keep = [users_tab, user_ids, items_tab, w, wu, c, ops._topk_ws_cache.get(items_tab.device)]
            keep.extend(self.__dict__.get("_seeds", {}).values())
            local = [] if self._local_own is None else [self._mask_local] + list(self._mask_local.__dict__.get("_row_ranges", {}).values())
            for csr in [self.mask] + list(self.mask.__dict__.get("_row_ranges", {}).values()) + local:
                keep.extend(csr.__dict__.get("_mask_bits", {}).values())
            entry = self._graphs[key] = (stages, out, keep, first_bufs)
        self._last_entry = entry
        (ga, local, gathered, gb), out = entry[0], entry[1]
        ga.replay()
        if gb is not None:
            nv, ni = sharding.gather_topk(local[0], local[1])
            gathered[0].copy_(nv); gathered[1].copy_(ni)
            gb.replay()
        return out

    # ------------------------------------------------------------------ c sweep (tuners)
    def _sweep_direct(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c_dev):
        sig_i = ops.branch_sigmoid(items_tab, w)
        sig_u = ops.branch_sigmoid(users_tab, wu, user_ids) if kind in (ops.SCORE_RUBI_BOTH, ops.SCORE_DIRECT_MINUS_BOTH) else None
        vals, idx = ops.score_topk_sweep(kind, users_tab, user_ids, items_tab, max(Ks), sig_u, sig_i, c_dev, self.mask, 0,
                                         filter=self.filter)
        return torch.stack([self._finish(flavour, vals[g:g + 1], idx[g:g + 1], Ks) for g in range(c_dev.numel())])

    def sweep_means(self, flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, cs):
        """Column means of the per-user metrics for every c of `cs`, (len(cs), ...).  On one GPU the values go through
        the shared-listing-pass kernel in groups of up to four (one captured graph per group size, the group's values
        in a device array the kernels read at run time); item-sharded runs evaluate c by c."""
        from . import _lib
        cs = [float(c) for c in cs]
        # The shared-listing-pass kernels follow the candidate filter (k_score_stream_bs under "bf16": 0.31 ms per value
        # on the Gowalla shape against 0.43 one evaluation at a time, tools/bench_sweep.py).  MACR_SWEEP_ONE_BY_ONE=1
        # sends every value through the seeded, graph-replayed single evaluation instead (A/B switch).
        one_by_one = os.environ.get("MACR_SWEEP_ONE_BY_ONE", "0") == "1"
        if (one_by_one or sharding.world()[1] > 1 or kind == ops.SCORE_NORMAL or self.n_queries > self.max_queries_per_pass
                or max(Ks) > _lib.MAX_TOPK_FUSED):           # (the shared-listing-pass kernels rank K <= 32)
            return torch.stack([self._means(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c).clone() for c in cs])
        outs = []
        for a in range(0, len(cs), _lib.MAX_SWEEP):
            chunk = cs[a:a + _lib.MAX_SWEEP]
            n = len(chunk)
            bufs = self.__dict__.setdefault("_c_sweep", {})
            if n not in bufs:
                bufs[n] = torch.zeros(n, dtype=torch.float32, device=self.device)
            c_dev = bufs[n]
            c_dev.copy_(torch.tensor(chunk, dtype=torch.float32), non_blocking=False)
            if not self.use_graph:
                outs.append(self._sweep_direct(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c_dev).clone())
                continue
            key = ("sweep", n, flavour, self.filter, kind, users_tab.data_ptr(), None if user_ids is None else user_ids.data_ptr(),
                   items_tab.data_ptr(), Ks, w.data_ptr(), None if wu is None else wu.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
            entry = self._graphs.get(key)
            if entry is None:
                self._sweep_direct(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c_dev)       # warm-up
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._sweep_direct(flavour, kind, users_tab, user_ids, items_tab, Ks, w, wu, c_dev)
                keep = [users_tab, user_ids, items_tab, w, wu, c_dev, ops._sweep_ws_cache.get(items_tab.device)]
                keep.extend(self.mask.__dict__.get("_mask_bits", {}).values())
                entry = self._graphs[key] = ((g, None, None, None), out, keep)
            entry[0][0].replay()
            outs.append(entry[1].clone())
        return torch.cat(outs)

    def test_mf_sweep(self, kind, users_tab, user_ids, items_tab, Ks, w, wu, cs):
        """test_mf for every c of `cs` -> list of result dicts (the c sweep of macr_mf/tune.py:545-578)."""
        m = self.sweep_means("mf", kind, users_tab, user_ids, items_tab, tuple(Ks), w, wu, cs).cpu().numpy()
        return [{'precision': x[0].copy(), 'recall': x[1].copy(), 'ndcg': x[2].copy(), 'hit_ratio': x[3].copy()} for x in m]

    def test_lgcn_sweep(self, kind, users_tab, user_ids, items_tab, Ks, w, wu, cs):
        top_show = np.sort(np.asarray(Ks))
        max_top = int(top_show.max())
        m = self.sweep_means("lgcn", kind, users_tab, user_ids, items_tab, tuple(Ks), w, wu, cs).cpu().numpy()
        out = []
        for x in m:
            final = x.reshape(5, max_top)[:, top_show - 1]
            out.append({'hr': final[2].copy(), 'recall': final[1].copy(), 'ndcg': final[3].copy()})
        return out

    # ------------------------------------------------------------------ LightGCN flavour
    def test_lgcn(self, kind, users_tab, user_ids, items_tab, Ks, w=None, wu=None, c=0.0):
        """-> {'hr','recall','ndcg'}: np.ndarray(len(Ks)) (batch_test.py:134-161): C++-style fp32 prefix
        metrics, HR := 1[recall@k != 0], mean over users, columns Ks-1 in ascending-K order."""
        top_show = np.sort(np.asarray(Ks))
        max_top = int(top_show.max())
        final = self._means("lgcn", kind, users_tab, user_ids, items_tab, tuple(Ks), w, wu, c).cpu().numpy()
        final = final.reshape(5, max_top)[:, top_show - 1]
        return {'hr': final[2].copy(), 'recall': final[1].copy(), 'ndcg': final[3].copy()}


def eval_score_matrix_foldout(score_matrix, test_items, top_k=20, thread_num=None):
    """Drop-in for macr_lightgcn/evaluator/cpp/evaluate_foldout.py:12-18 backed by the HIP kernels
    (macr_topk_scores + macr_metrics_foldout).  score_matrix: (U,N) array-like with train items already
    at -inf (batch_test.py:129); test_items: list of per-user ground-truth id lists.
    Returns np.float32 (U, 5*top_k) laid out [precision|recall|ap|ndcg|mrr].  thread_num is accepted and
    ignored.  Ties rank by ascending item id (the C++ leaves tie order to std::partial_sort_copy)."""
    if len(score_matrix) != len(test_items):
        raise ValueError("The lengths of score_matrix and test_items are not equal.")
    dev = torch.device("cuda", torch.cuda.current_device())
    if isinstance(score_matrix, torch.Tensor):
        scores = score_matrix.to(device=dev, dtype=torch.float32).contiguous()
    else:
        scores = torch.from_numpy(np.ascontiguousarray(score_matrix, dtype=np.float32)).to(dev)
    idx, _ = ops.topk_scores(scores, top_k, want_vals=False)
    gt = ops.CSR.from_lists(test_items, dev)
    return ops.metrics_foldout(idx, gt).cpu().numpy()
