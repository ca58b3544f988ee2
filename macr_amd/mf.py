"""MF family host model: the `BPRMF` object the MF CLI drives, on the HIP path.

Mirrors the part of macr_mf/model.py::BPRMF (:13-326) that the README commands
exercise:
    --train normalbce   -> opt_bce / loss_bce / mf_loss_bce / reg_loss_bce           (:92-95, :277-287)
    --train rubibceboth -> opt_two_bce_both / loss_two_bce_both / ...                (:71-74, :185-222)
    --train rubibce     -> opt_two_bce / loss_two_bce / ...  (item branch only)      (:67-69, :158-183)
    --test  normal      -> batch_ratings                                             (:45)
    --test  rubi        -> rubi_ratings_both (rubibceboth) | rubi_ratings (other losses) + update_c   (:199, :141, :313)
    direct_minus_ratings(_both) (:142, :201) are served too (test(model_type="direct_minus_c"))
The reference builds a TF1 graph and the CLI talks to it through
`sess.run(fetches, feed_dict)`; here the same attribute names are plain fetch
handles and `Session.run` dispatches them to the C-ABI kernels, so a caller
written against the reference keeps working.  The fast path (`train_step`,
`Evaluator`) avoids the per-step host synchronisation `sess.run` implies.

Everything else in model.py (bpr / rubi / userc losses, BIASMF,
IPS_BPRMF, CausalE) is out of scope and raises NotImplementedError.
"""
import math

import numpy as np
import torch

from . import ops


class Fetch(object):
    """Symbolic handle standing where the reference has a tf.Tensor / tf.Operation."""
    __slots__ = ("name", "role", "kind")

    def __init__(self, name, role, kind=None):
        self.name, self.role, self.kind = name, role, kind

    def __repr__(self):
        return "<macr fetch %s>" % self.name


def xavier_uniform(shape, generator, device):
    """tf.contrib.layers.xavier_initializer(): U(-L, L), L = sqrt(6/(fan_in+fan_out)) with
    fan_in=rows, fan_out=cols for a 2-D shape.  TF's Philox stream cannot be
    replayed; values come from a torch.Generator (parity runs inject weights)."""
    limit = math.sqrt(6.0 / (shape[0] + shape[1]))
    t = (torch.rand(shape, generator=generator, dtype=torch.float32) * 2.0 - 1.0) * limit
    return t.to(device)


class BPRMF(object):
    _TRAIN = {"normalbce": ("bce", ops.LOSS_NORMALBCE), "rubibceboth": ("two_bce_both", ops.LOSS_RUBIBCEBOTH),
              "rubibce": ("two_bce", ops.LOSS_RUBIBCE)}

    def __init__(self, args, data_config, device=None, seed=12345, weights=None):
        self.n_users = data_config['n_users']
        self.n_items = data_config['n_items']
        self.decay = args.regs
        self.emb_dim = args.embed_size
        self.lr = args.lr
        self.batch_size = args.batch_size
        self.verbose = getattr(args, "verbose", 0)
        self.c = args.c
        self.alpha = args.alpha
        self.beta = args.beta
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        # placeholders (model.py:27-29)
        self.users = Fetch("users", "placeholder")
        self.pos_items = Fetch("pos_items", "placeholder")
        self.neg_items = Fetch("neg_items", "placeholder")
        # parameters (model.py:107-122, :59-60); rubi_c starts at 0 (:117)
        self.weights = self.init_weights(seed, weights)
        self.rubi_c = 0.0
        hyper = ops.make_hyper(self.lr, self.decay, self.alpha, self.beta, self.batch_size)
        # one optimizer instance (own Adam slots and step count) per `minimize` call of the reference
        self._opt = {}
        for train, (suffix, kind) in self._TRAIN.items():
            self._opt[kind] = ops.MFState(self.weights['user_embedding'], self.weights['item_embedding'],
                                          self.w, self.w_user, hyper, self.batch_size)
            setattr(self, "opt_" + suffix, Fetch("opt_" + suffix, "opt", kind))
            setattr(self, "loss_" + suffix, Fetch("loss_" + suffix, "loss", kind))
            setattr(self, "mf_loss_" + suffix, Fetch("mf_loss_" + suffix, "mf_loss", kind))
            setattr(self, "reg_loss_" + suffix, Fetch("reg_loss_" + suffix, "reg_loss", kind))
        # the MFState objects must all alias the same parameter storage
        st0 = self._opt[ops.LOSS_NORMALBCE]
        for st in self._opt.values():
            st.P, st.Q, st.w, st.wu = st0.P, st0.Q, st0.w, st0.wu
        self.weights['user_embedding'], self.weights['item_embedding'] = st0.P, st0.Q
        self.w, self.w_user = st0.w, st0.wu
        # inference handles
        self.batch_ratings = Fetch("batch_ratings", "ratings", ops.SCORE_NORMAL)
        self.rubi_ratings_both = Fetch("rubi_ratings_both", "ratings", ops.SCORE_RUBI_BOTH)
        self.rubi_ratings = Fetch("rubi_ratings", "ratings", ops.SCORE_RUBI)                               # :141
        self.direct_minus_ratings = Fetch("direct_minus_ratings", "ratings", ops.SCORE_DIRECT_MINUS)      # :142
        self.direct_minus_ratings_both = Fetch("direct_minus_ratings_both", "ratings", ops.SCORE_DIRECT_MINUS_BOTH)   # :201
        for name in ("opt", "opt_two", "opt2", "opt2_bce", "opt3", "opt3_bce", "opt_userc_bce",
                     "user_const_ratings", "item_const_ratings", "user_rand_ratings", "item_rand_ratings",
                     "rubi_ratings_userc", "rubi_ratings_both_poptest"):
            setattr(self, name, Fetch(name, "unsupported"))
        self._statistics_params()

    def init_weights(self, seed, weights=None):
        gen = torch.Generator().manual_seed(seed)
        dev, d = self.device, self.emb_dim
        # an --embed_size outside {32,64,128,256} runs at the next supported width with zero columns (ops.padded_dim:
        # exact); the Xavier limits are those of the d-wide shapes
        dp = self.d_pad = ops.padded_dim(d)
        out = dict()
        if weights is not None:      # injected (parity runs): numpy/torch arrays
            as_t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(dev).contiguous()
            out['user_embedding'] = ops.pad_cols(as_t(weights['user_embedding']), dp)
            out['item_embedding'] = ops.pad_cols(as_t(weights['item_embedding']), dp)
            self.w = ops.pad_cols(as_t(weights['w']).reshape(-1), dp)
            self.w_user = ops.pad_cols(as_t(weights['w_user']).reshape(-1), dp)
        else:
            out['user_embedding'] = ops.pad_cols(xavier_uniform((self.n_users, d), gen, dev), dp)
            out['item_embedding'] = ops.pad_cols(xavier_uniform((self.n_items, d), gen, dev), dp)
            self.w = ops.pad_cols(xavier_uniform((d, 1), gen, dev).reshape(-1), dp)
            self.w_user = ops.pad_cols(xavier_uniform((d, 1), gen, dev).reshape(-1), dp)
        return out

    def _statistics_params(self):
        total = (self.n_users + self.n_items) * self.emb_dim
        if self.verbose > 0:
            print("#params: %d" % total)

    # ------------------------------------------------------------------ reference API
    def update_c(self, sess, c):
        """model.py:313 -- sess is accepted and ignored."""
        self.rubi_c = float(c)

    # ------------------------------------------------------------------ fast path
    def kind_of(self, train):
        if train not in self._TRAIN:
            raise NotImplementedError("--train %s is not on the MI355X hot path (normalbce | rubibce | rubibceboth)" % train)
        return self._TRAIN[train][1]

    def to_device_batch(self, users, pos_items, neg_items):
        """Python lists (what Data.sample returns) -> one (3,B) int32 device tensor."""
        arr = np.asarray([users, pos_items, neg_items], dtype=np.int32)
        # (the library orders the batch by positive item itself, on the device: batch_sort_block in train_kernels.hip)
        host = torch.from_numpy(np.ascontiguousarray(arr)).pin_memory()
        return host.to(self.device, non_blocking=True)

    def train_step(self, kind, batch, losses=None, defer=False):
        """batch: (3,B) int32 device tensor.  Returns the (3,) device tensor {loss, mf_loss, reg_loss};
        no host synchronisation.  defer=True lets consecutive rubibceboth steps overlap the dense Adam pass of
        one step with the (B,B) kernel of the next (MACR_STEP_DEFER); the parameters are then only up to date
        after sync(), which every reader below calls."""
        for k, st in self._opt.items():
            if k != kind:
                st.flush()
        return self._opt[kind].step(kind, batch[0], batch[1], batch[2], losses, defer=defer)

    def sync(self):
        """Complete any pending parameter update (stream-ordered; no host synchronisation)."""
        for st in self._opt.values():
            st.flush()

    def opt_state(self, kind):
        return self._opt[kind]

    @property
    def user_embedding(self):
        return self.weights['user_embedding']

    @property
    def item_embedding(self):
        return self.weights['item_embedding']

    def ratings(self, kind, user_batch):
        """Dense (U,N) score matrix: the literal sess.run(model.batch_ratings|rubi_ratings_both)."""
        self.sync()
        uid = torch.as_tensor(list(user_batch), dtype=torch.int32, device=self.device)
        sig_u = sig_i = None
        if kind != ops.SCORE_NORMAL:
            sig_i = ops.branch_sigmoid(self.item_embedding, self.w)
        if kind in (ops.SCORE_RUBI_BOTH, ops.SCORE_DIRECT_MINUS_BOTH):
            sig_u = ops.branch_sigmoid(self.user_embedding, self.w_user, uid)
        return ops.score_matrix(kind, self.user_embedding, uid, self.item_embedding, sig_u, sig_i, self.rubi_c)

    def parameters(self):
        self.sync()
        return [self.user_embedding, self.item_embedding, self.w, self.w_user]

    def state_dict(self):
        self.sync()
        sd = {"user_embedding": self.user_embedding, "item_embedding": self.item_embedding, "w": self.w,
              "w_user": self.w_user, "rubi_c": self.rubi_c}
        for kind, st in self._opt.items():
            for name in ("mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu", "adam_pow"):
                sd["opt%d.%s" % (kind, name)] = getattr(st, name)
        return sd

    def load_state_dict(self, sd):
        self.sync()
        self.user_embedding.copy_(sd["user_embedding"]); self.item_embedding.copy_(sd["item_embedding"])
        self.w.copy_(sd["w"]); self.w_user.copy_(sd["w_user"]); self.rubi_c = float(sd["rubi_c"])
        for kind, st in self._opt.items():
            fresh = "opt%d.adam_pow" % kind not in sd          # (a row-sharded run's checkpoint carries ITS optimizer only)
            for name in ("mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu"):
                getattr(st, name).zero_() if fresh else getattr(st, name).copy_(sd["opt%d.%s" % (kind, name)])
            st.adam_pow.copy_(torch.tensor([st.hyper.beta1, st.hyper.beta2]) if fresh else sd["opt%d.adam_pow" % kind])


class ShardedBPRMF(object):
    """BPRMF whose embedding tables, Adam slots and gradient scratch are ROW-SHARDED over the ranks of the process group
    (`--row_shard 1`; macr_amd/sharded_train.py): what the MF CLI needs of BPRMF, same names.  The reference keeps each table
    in one tf.Variable (macr_mf/model.py:112-113); at BASELINE configs[4] sizes the TF-style dense Adam pass over every row
    is what a step costs, and it divides by the number of ranks.  Every rank feeds the SAME batch (same sampler seed).
    The initial tables are the ones BPRMF(seed) draws (the same host generator), so a sharded and an unsharded run start
    from the same model."""
    sharded = True
    _TRAIN = BPRMF._TRAIN

    def __init__(self, args, data_config, device=None, seed=12345, layout="interleaved"):
        from . import sharded_train, sharding
        self.n_users, self.n_items = data_config['n_users'], data_config['n_items']
        self.decay, self.emb_dim, self.lr, self.batch_size = args.regs, args.embed_size, args.lr, args.batch_size
        self.c, self.alpha, self.beta = args.c, args.alpha, args.beta
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rank, self.world = sharding.world()
        self.rubi_c = 0.0
        gen = torch.Generator().manual_seed(seed)
        d = self.emb_dim
        dp = self.d_pad = ops.padded_dim(d)
        cpu = torch.device("cpu")
        self.own_u = sharded_train.Owned(self.n_users, self.rank, self.world, layout)
        self.own_i = sharded_train.Owned(self.n_items, self.rank, self.world, layout)
        # the draws of BPRMF.init_weights, in its order; only this rank's rows go to the device
        P = self.own_u.take(xavier_uniform((self.n_users, d), gen, cpu)).to(self.device)
        Q = self.own_i.take(xavier_uniform((self.n_items, d), gen, cpu)).to(self.device)
        w = ops.pad_cols(xavier_uniform((d, 1), gen, cpu).reshape(-1), dp).to(self.device)
        wu = ops.pad_cols(xavier_uniform((d, 1), gen, cpu).reshape(-1), dp).to(self.device)
        self._hyper = ops.make_hyper(self.lr, self.decay, self.alpha, self.beta, self.batch_size)
        self._init = (ops.pad_cols(P, dp), ops.pad_cols(Q, dp), w, wu)
        self._layout = layout
        self._models = {}
        self._q = {}

    def kind_of(self, train):
        if train not in self._TRAIN:
            raise NotImplementedError("--train %s is not on the MI355X hot path (normalbce | rubibce | rubibceboth)" % train)
        return self._TRAIN[train][1]

    def _model(self, kind):
        from . import sharded_train
        if kind not in self._models:
            if self._models:
                raise NotImplementedError("a row-sharded model trains with one loss kind per run")
            P, Q, w, wu = self._init
            self._models[kind] = sharded_train.RowShardedMF(
                None, None, w, wu, sharded_train.HipBackend(kind, self.d_pad, self._hyper, self.device), rank=self.rank,
                world=self.world, shards=(P, Q, self.n_users, self.n_items), layout=self._layout)
        return self._models[kind]

    def _any(self):
        """the one model of this run: created with the loss kind the CLI trains with (`default_kind`, set from --train)
        when a reader -- an evaluation, a checkpoint -- comes before the first step"""
        if not self._models:
            kind = getattr(self, "default_kind", None)
            self._model(ops.LOSS_RUBIBCEBOTH if kind is None else kind)
        return next(iter(self._models.values()))

    def to_device_batch(self, users, pos_items, neg_items):
        """as BPRMF.to_device_batch; the host copy is kept for the step that consumes it (routing table of the split step
        without a device synchronisation: sharded_train.RowShardedMF.route_counts_host)"""
        arr = np.asarray([users, pos_items, neg_items], dtype=np.int32)
        dev = torch.from_numpy(np.ascontiguousarray(arr)).pin_memory().to(self.device, non_blocking=True)
        # the host copy travels WITH the tensor object (an attribute), not keyed by its address: the caching allocator hands the
        # same address to the next (3,B) tensor, and a batch from another source must never meet a stale host copy
        dev._macr_host_batch = arr
        return dev

    def update_c(self, sess, c):
        self.rubi_c = float(c)

    def train_step(self, kind, batch, losses=None, defer=False):
        m = self._model(kind)
        counts = None
        hb = getattr(batch, "_macr_host_batch", None)        # set by to_device_batch on THIS tensor object only
        if hb is not None and hb.shape == tuple(batch.shape) and self.world > 1 and m.split and kind != ops.LOSS_NORMALBCE:
            counts = m.route_counts_host(hb[0], hb[1], hb[2])
        out = m.step(batch[0], batch[1], batch[2], counts=counts)
        if losses is not None:
            losses.copy_(out)
        return out

    def sync(self):
        pass

    def parameters(self):
        return []                                   # nothing is replicated but w, w_user, which are bit-identical by construction

    @property
    def user_embedding(self):
        return self._any().P                        # THIS RANK'S rows (own_u)

    @property
    def item_embedding(self):
        return self._any().Q                        # THIS RANK'S rows (own_i): the evaluator's item shard

    @property
    def w(self):
        return self._any().w

    @property
    def w_user(self):
        return self._any().wu

    def query_rows(self, uid):
        """(U, d) rows of the query users on every rank: each rank adds the rows it owns, one all-reduce (a persistent
        buffer per query set: the evaluator replays a captured graph that reads it)"""
        key = uid.data_ptr()
        if key not in self._q:
            self._q[key] = torch.zeros((uid.numel(), self.d_pad), dtype=torch.float32, device=self.device)
        buf, m = self._q[key], self._any()
        own, loc = self.own_u.local_of(uid.long())
        buf.zero_()
        buf[own] = m.rows("P", loc[own])           # (as of the current step; the user table itself is not brought up to date)
        m._all_reduce(buf, "query_rows")
        return buf

    def _full(self, local, own):
        m = self._any()
        if self.world == 1:
            return local.clone()
        full = torch.zeros((own.n_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        full[own.global_ids(local.device)] = local
        m._all_reduce(full, "full_table")
        return full

    def state_dict(self):
        """The model as BPRMF saves it (full tables): COLLECTIVE -- every rank calls it, the main rank writes the file."""
        sd = {"rubi_c": self.rubi_c}
        for kind, m in self._models.items():
            sd["user_embedding"], sd["item_embedding"] = self._full(m.P, self.own_u), self._full(m.Q, self.own_i)
            sd["w"], sd["w_user"] = m.w, m.wu
            for name, own in (("mP", self.own_u), ("vP", self.own_u), ("mQ", self.own_i), ("vQ", self.own_i)):
                sd["opt%d.%s" % (kind, name)] = self._full(getattr(m, name), own)
            for name in ("mw", "vw", "mwu", "vwu"):
                sd["opt%d.%s" % (kind, name)] = getattr(m, name)
            sd["opt%d.adam_pow" % kind] = m.backend.adam_pow
            sd["row_shard_kind"] = kind
        return sd

    def load_state_dict(self, sd):
        """`sd` holds FULL tables (what state_dict() / BPRMF.state_dict() write); every rank takes its own rows.  NOT
        collective.  The loss kind is the checkpoint's (`row_shard_kind`), else the one the CLI trains with, else the one
        this model already runs; a checkpoint without that optimizer's slots (an unsharded `--pretrain` file of another
        loss, a weights-only file) starts that optimizer fresh."""
        if self._models:
            kind = next(iter(self._models))
        elif "row_shard_kind" in sd:
            kind = int(sd["row_shard_kind"])
        else:
            kind = getattr(self, "default_kind", None)
            if kind is None:
                have = sorted(int(k[3:].split(".")[0]) for k in sd if k.startswith("opt") and k.endswith(".adam_pow"))
                kind = have[0] if len(have) == 1 else ops.LOSS_RUBIBCEBOTH
        m = self._model(kind)
        dev = self.device
        m.P.copy_(self.own_u.take(sd["user_embedding"].to(dev))); m.Q.copy_(self.own_i.take(sd["item_embedding"].to(dev)))
        m.w.copy_(sd["w"]); m.wu.copy_(sd["w_user"]); self.rubi_c = float(sd["rubi_c"])
        if "opt%d.adam_pow" % kind not in sd:
            for name in ("mP", "vP", "mQ", "vQ", "mw", "vw", "mwu", "vwu"):
                getattr(m, name).zero_()
            h = m.backend.hyper
            m.backend.adam_pow.copy_(torch.tensor([h.beta1, h.beta2]))
            return
        for name, own in (("mP", self.own_u), ("vP", self.own_u), ("mQ", self.own_i), ("vQ", self.own_i)):
            getattr(m, name).copy_(own.take(sd["opt%d.%s" % (kind, name)].to(dev)))
        for name in ("mw", "vw", "mwu", "vwu"):
            getattr(m, name).copy_(sd["opt%d.%s" % (kind, name)])
        m.backend.adam_pow.copy_(sd["opt%d.adam_pow" % kind])


class Session(object):
    """Stands where the reference has tf.Session: `run(fetches, feed_dict)` with the fetch handles of
    BPRMF / LightGCN.  Training fetch lists return [None, loss, mf_loss, reg_loss] as Python floats
    (one device->host sync per call, like the reference); a ratings fetch returns the (U,N) matrix."""

    def __init__(self, model=None):
        self.model = model

    def bind(self, model):
        self.model = model
        return self

    def run(self, fetches, feed_dict=None):
        m = self.model
        single = not isinstance(fetches, (list, tuple))
        flist = [fetches] if single else list(fetches)
        feed = feed_dict or {}
        roles = [f.role for f in flist]
        if "unsupported" in roles:
            bad = [f.name for f in flist if f.role == "unsupported"]
            raise NotImplementedError("fetch %s is outside the MI355X hot path" % bad)
        out = [None] * len(flist)
        if "opt" in roles:
            kind = flist[roles.index("opt")].kind
            batch = m.to_device_batch(feed[m.users], feed[m.pos_items], feed[m.neg_items])
            losses = m.train_step(kind, batch).cpu().numpy()
            names = getattr(m, "_loss_slots", {"loss": 0, "mf_loss": 1, "reg_loss": 2})
            for k, f in enumerate(flist):
                if f.role in names:
                    out[k] = losses[names[f.role]]
                elif f.role == "zero":
                    out[k] = np.zeros(1, np.float32)
        elif "ratings" in roles:
            for k, f in enumerate(flist):
                out[k] = m.ratings(f.kind, feed[m.users]).cpu().numpy()
        else:
            raise NotImplementedError("nothing to run in %r" % (flist,))
        return out[0] if single else out
