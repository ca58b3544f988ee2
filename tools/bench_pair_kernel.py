#!/usr/bin/env python3
"""Roofline of the fused per-pair kernels at synthetic scale (SURVEY.md 8d caveat: at the configured batch sizes the
pair kernels move 6-13 MB and are launch/latency bound, so the HBM roofline fraction is shown at B = 2^18..2^22 on
large tables).  Algorithmic bytes per triple: 24*d+12 (3 rows read, 3 gradient rows written, indices)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from macr_amd import ops, synth

dev = torch.device("cuda"); d = 64
n_users, n_items = 4_000_000, 1_000_000
gen = torch.Generator(device=dev).manual_seed(1)
P = synth.xavier_table(n_users, d, gen, dev); Q = synth.xavier_table(n_items, d, gen, dev)
w = synth.xavier_table(d, 1, gen, dev).reshape(-1); wu = synth.xavier_table(d, 1, gen, dev).reshape(-1)
out = []
for logB in ([int(a) for a in sys.argv[1:]] or (12, 14, 16, 18, 20)):       # argv: only these log2(B) (profiling runs)
    B = 1 << logB
    for kind, name in ((ops.LOSS_NORMALBCE, "normalbce"),):
        state = ops.MFState(P, Q, w, wu, ops.make_hyper(1e-3, 1e-5, 1e-2, 1e-3, B), B)
        u = torch.randperm(n_users, generator=gen, device=dev)[:B].to(torch.int32)
        i = torch.randint(0, n_items, (B,), generator=gen, device=dev).to(torch.int32)
        j = torch.randint(0, n_items, (B,), generator=gen, device=dev).to(torch.int32)
        for _ in range(2):
            state.step(kind, u, i, j)
        ops.timing_begin()
        for _ in range(5):
            state.step(kind, u, i, j)
        agg = {}
        for n, ms in ops.timing_end(64):
            a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
        us = {k: 1e3 * v[1] / v[0] for k, v in agg.items()}
        pair_us = us.get("pair_normal", 0.0)
        bytes_pair = B * (24 * d + 12)
        bytes_adam = 24 * d * (n_users + n_items)
        # (B > 8192: "seg_index" + "adam_indexed" when the Adam pass sums the staged rows itself, "seg_reduce" +
        #  "adam_dense" under MACR_SEG_UNFUSED=1)
        grad_us = pair_us + us.get("ref_sort", 0.0) + us.get("seg_reduce", 0.0) + us.get("seg_index", 0.0) + us.get("batch_sort", 0.0)
        adam_us = us.get("adam_dense", 0.0) + us.get("adam_indexed", 0.0)
        out.append({"B": B, "loss": name, "kernels_us": {k: round(v, 1) for k, v in us.items()},
                    "pair_GBps": bytes_pair / (pair_us * 1e-6) / 1e9, "pair_frac_of_8TBps": bytes_pair / (pair_us * 1e-6) / 8e12,
                    "gradient_path_us": round(grad_us, 1),          # pair kernel + reference sort + segment reduce
                    "gradient_path_frac_of_8TBps": bytes_pair / (grad_us * 1e-6) / 8e12,
                    "adam_GBps": bytes_adam / (adam_us * 1e-6) / 1e9,
                    "step_us": round(sum(us.values()), 1)})
        del state
print(json.dumps(out))
