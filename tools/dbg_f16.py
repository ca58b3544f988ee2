import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import oracle
from macr_amd import ops
sys.path.insert(0, '/root/repo/tests')
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def random_mask(rs, U, N, m, heavy=()):
    out=[]
    for u in range(U):
        k = m*20 if u in heavy else m
        out.append(sorted(rs.choice(N, min(k,N-1), replace=False).tolist()))
    return out
for d in (64,):
  for variant in ("test", "sigu1", "sigi_mild", "c0"):
    kind = oracle.SCORE_RUBI_BOTH
    rs = np.random.RandomState(4000 + 17 * kind + d)
    U, N, K = 600, 5000, 20
    P = (rs.standard_normal((U, d)) * 0.4).astype(np.float32)
    Q = (rs.standard_normal((N, d)) * 0.4).astype(np.float32)
    sig_u = (1.0 / (1.0 + np.exp(-rs.standard_normal(U) * 2.0))).astype(np.float32)
    sig_i = (1.0 / (1.0 + np.exp(-rs.standard_normal(N) * 2.0))).astype(np.float32)
    sig_u[:5] = [1.0, 1e-2, 0.03, 0.1, 0.5]
    sig_i[:5] = [1.0, 1e-4, 1e-8, 1e-12, 0.25]
    c = 40.0
    if variant == "sigu1": sig_u[:] = 1.0
    if variant == "sigi_mild": sig_i[:5] = 0.5
    if variant == "c0": c = 0.0
    mask = random_mask(rs, U, N, 30, heavy=(7,))
    mcsr = ops.CSR.from_lists(mask, "cuda")
    wv, wi, _ = oracle.score_topk(kind, P, Q, K, sig_u, sig_i, c, oracle.csr_from_lists(mask))
    for filt in ("bf16", "f16"):
        ops.set_eval_filter(filt)
        stats = torch.zeros(2, dtype=torch.int32, device="cuda")
        v, ix = ops.score_topk(kind, dev(P), None, dev(Q), K, dev(sig_u), dev(sig_i), c, mcsr, stats=stats)
        ok = np.array_equal(ix[0].cpu().numpy(), wi)
        # margins
        sc, mg = (ops.test_f16_scores if filt=="f16" else ops.test_bf16_scores)(kind, dev(P), dev(Q), dev(sig_u), dev(sig_i), c)
        sc = sc.cpu().numpy(); mg = mg.cpu().numpy()
        srt = -np.sort(-sc, axis=1)
        gap = srt[:, K-1] - srt[:, 63]
        tau160 = srt[:, 159]
        n_listed = (sc >= (tau160 - 1.01*mg)[:, None]).sum(1)
        print(variant, filt, "stats", stats.cpu().numpy().tolist(), "ok", ok, "margin med %.3g max %.3g" % (np.median(mg), mg.max()),
              "gap K..64 med %.3g min %.3g" % (np.median(gap), gap.min()), "frac gap<2.02m %.3f" % (gap < 2.02*mg).mean(),
              "listed med %d max %d" % (np.median(n_listed), n_listed.max()))
