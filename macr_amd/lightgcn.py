"""LightGCN host model on the HIP path.

Mirrors the part of macr_lightgcn/LightGCN.py::LightGCN (:32-555) the README
commands exercise:
    --alg_type lightgcn, --adj_type pre
    --loss bce     -> opt_bce / loss_bce / mf_loss_bce / emb_loss_bce / reg_loss_bce            (:180-186, :415-429)
    --loss bceboth -> opt_two_bce_both / loss_two_bce_both / ...                               (:196-201, :495-532)
    --test normal  -> batch_ratings (:166);  --test rubiboth -> rubi_ratings_both (:509) + update_c (:554)
ngcf / gcn / gcmc embeddings, bpr / bce1 / bce2 losses, node and message
dropout, pretrained restore are out of scope (NotImplementedError).
"""
import ast

import numpy as np
import torch

from . import ops
from .mf import Fetch, xavier_uniform


class LightGCN(object):
    _LOSS = {"bce": ("bce", ops.LOSS_NORMALBCE), "bceboth": ("two_bce_both", ops.LOSS_RUBIBCEBOTH)}

    def __init__(self, data_config, args, pretrain_data=None, device=None, seed=12345, weights=None):
        if pretrain_data is not None:
            raise NotImplementedError("pretrained restore is out of scope")
        if getattr(args, "alg_type", "lightgcn") != "lightgcn":
            raise NotImplementedError("--alg_type %s is out of scope (lightgcn only)" % args.alg_type)
        if getattr(args, "node_dropout_flag", 0):
            raise NotImplementedError("node dropout is out of scope")
        self.model_type = 'LightGCN'
        self.adj_type = args.adj_type
        self.alg_type = args.alg_type
        self.n_users = data_config['n_users']
        self.n_items = data_config['n_items']
        self.norm_adj = data_config['norm_adj']
        # the backward pass through the propagation runs on the TRANSPOSED operator (dE0 = A^T dE): A itself for the symmetric
        # `pre` and `plain` matrices, a second CSR (with an SpMM plan of its own) for the row-normalised norm / gcmc / mean
        # ones, D^-1 A (LightGCN.py:667-678; macr_lgcn_train_step_t)
        asym = abs(self.norm_adj - self.norm_adj.T)
        self.asymmetric = bool(asym.nnz and asym.max() > 1e-6 * abs(self.norm_adj).max())
        self.n_nonzero_elems = self.norm_adj.count_nonzero()
        self.lr = args.lr
        self.emb_dim = args.embed_size
        self.batch_size = args.batch_size
        self.weight_size = ast.literal_eval(args.layer_size)
        self.n_layers = len(self.weight_size)
        self.regs = ast.literal_eval(args.regs)
        self.decay = self.regs[0]
        self.verbose = args.verbose
        self.Ks = ast.literal_eval(args.Ks)
        self.alpha, self.beta = args.alpha, args.beta
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.log_dir = self.create_model_str(args)
        self.users = Fetch("users", "placeholder")
        self.pos_items = Fetch("pos_items", "placeholder")
        self.neg_items = Fetch("neg_items", "placeholder")
        self.node_dropout = Fetch("node_dropout", "placeholder")
        self.mess_dropout = Fetch("mess_dropout", "placeholder")
        # parameters (:221-254): ego table T = [user_embedding ; item_embedding], branch vectors
        gen = torch.Generator().manual_seed(seed)
        d = self.emb_dim
        # an --embed_size outside {32,64,128,256} runs at the next supported width with zero columns (ops.padded_dim:
        # exact -- the propagation is linear, zero columns stay zero); the Xavier limits are those of the d-wide shapes
        dp = self.d_pad = ops.padded_dim(d)
        if weights is not None:
            as_t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(device).contiguous()
            T = torch.cat([as_t(weights['user_embedding']), as_t(weights['item_embedding'])]).contiguous()
            w, wu = as_t(weights['w']).reshape(-1), as_t(weights['w_user']).reshape(-1)
        else:
            T = torch.cat([xavier_uniform((self.n_users, d), gen, device),
                           xavier_uniform((self.n_items, d), gen, device)]).contiguous()
            w = xavier_uniform((d, 1), gen, device).reshape(-1)
            wu = xavier_uniform((d, 1), gen, device).reshape(-1)
        T, w, wu = ops.pad_cols(T, dp), ops.pad_cols(w, dp), ops.pad_cols(wu, dp)
        self.rubi_c = 0.0
        adj = ops.CSR.from_scipy(self.norm_adj, device)
        adj_t = None
        if self.asymmetric:
            at = self.norm_adj.T.tocsr()
            at.sort_indices()
            adj_t = ops.CSR.from_scipy(at, device)
        hyper = ops.make_hyper(self.lr, self.decay, self.alpha, self.beta, self.batch_size)
        self._opt = {}
        for loss, (suffix, kind) in self._LOSS.items():
            self._opt[kind] = ops.LGCNState(T, self.n_users, self.n_items, w, wu, adj, self.n_layers, hyper,
                                            self.batch_size, adj_t=adj_t)
            setattr(self, "opt_" + suffix, Fetch("opt_" + suffix, "opt", kind))
            setattr(self, "loss_" + suffix, Fetch("loss_" + suffix, "loss", kind))
            setattr(self, "mf_loss_" + suffix, Fetch("mf_loss_" + suffix, "mf_loss", kind))
            setattr(self, "emb_loss_" + suffix, Fetch("emb_loss_" + suffix, "reg_loss", kind))
            setattr(self, "reg_loss_" + suffix, Fetch("reg_loss_" + suffix, "zero", kind))   # tf.constant(0.) :427
        st0 = self._opt[ops.LOSS_NORMALBCE]
        for st in self._opt.values():
            st.T, st.w, st.wu = st0.T, st0.w, st0.wu
        self.T, self.w, self.w_user = st0.T, st0.w, st0.wu
        self.weights = {'user_embedding': self.T[:self.n_users], 'item_embedding': self.T[self.n_users:]}
        self.batch_ratings = Fetch("batch_ratings", "ratings", ops.SCORE_NORMAL)
        self.rubi_ratings_both = Fetch("rubi_ratings_both", "ratings", ops.SCORE_RUBI_BOTH)
        for name in ("opt", "opt_two_bce1", "opt_two_bce2", "rubi_ratings1", "rubi_ratings2",
                     "batch_ratings_causal_c"):
            setattr(self, name, Fetch(name, "unsupported"))

    def create_model_str(self, args):
        log_dir = '/' + self.alg_type + '/layers_' + str(self.n_layers) + '/dim_' + str(self.emb_dim)
        log_dir += '/' + args.dataset + '/lr_' + str(self.lr) + '/reg_' + str(self.decay)
        return log_dir

    def update_c(self, sess, c):
        """LightGCN.py:554 (the second definition wins over :217)."""
        self.rubi_c = float(c)

    def kind_of(self, loss):
        if loss not in self._LOSS:
            raise NotImplementedError("--loss %s is not on the MI355X hot path (bce | bceboth)" % loss)
        return self._LOSS[loss][1]

    def to_device_batch(self, users, pos_items, neg_items):
        arr = np.asarray([users, pos_items, neg_items], dtype=np.int32)
        # (the library orders the batch by positive item itself, on the device: batch_sort_block in train_kernels.hip)
        host = torch.from_numpy(np.ascontiguousarray(arr)).pin_memory()
        return host.to(self.device, non_blocking=True)

    def train_step(self, kind, batch, losses=None, loss_only=False):
        """-> (3,) device tensor {loss, mf_loss, emb_loss}; invalidates the cached propagated table.
        loss_only=True: the reference's "test loss" pass (LightGCN.py:799-819) -- losses of the batch, no update."""
        if not loss_only:
            for st in self._opt.values():
                st._E = None
        return self._opt[kind].step(kind, batch[0], batch[1], batch[2], losses, loss_only=loss_only)

    def opt_state(self, kind):
        return self._opt[kind]

    def propagated(self):
        """(ua_embeddings, ia_embeddings) = split(mean(E0, A E0, A^2 E0 ...))  (:288-309); computed once
        per evaluation instead of once per user batch."""
        E = self._opt[ops.LOSS_NORMALBCE].propagated()
        return E[:self.n_users], E[self.n_users:]

    def ratings(self, kind, user_batch):
        uid = torch.as_tensor(list(user_batch), dtype=torch.int32, device=self.device)
        ua, ia = self.propagated()
        ia = ia.contiguous()
        sig_u = sig_i = None
        if kind != ops.SCORE_NORMAL:
            sig_i = ops.branch_sigmoid(ia, self.w)
        if kind in (ops.SCORE_RUBI_BOTH, ops.SCORE_DIRECT_MINUS_BOTH):
            sig_u = ops.branch_sigmoid(ua, self.w_user, uid)
        return ops.score_matrix(kind, ua, uid, ia, sig_u, sig_i, self.rubi_c)

    def state_dict(self):
        sd = {"T": self.T, "w": self.w, "w_user": self.w_user, "rubi_c": self.rubi_c}
        for kind, st in self._opt.items():
            for name in ("mT", "vT", "mw", "vw", "mwu", "vwu", "adam_pow"):
                sd["opt%d.%s" % (kind, name)] = getattr(st, name)
        return sd

    def load_state_dict(self, sd):
        self.T.copy_(sd["T"]); self.w.copy_(sd["w"]); self.w_user.copy_(sd["w_user"]); self.rubi_c = float(sd["rubi_c"])
        for kind, st in self._opt.items():
            st._E = None
            for name in ("mT", "vT", "mw", "vw", "mwu", "vwu", "adam_pow"):
                getattr(st, name).copy_(sd["opt%d.%s" % (kind, name)])

    def parameters(self):
        return [self.T, self.w, self.w_user]
