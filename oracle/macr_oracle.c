/* ============================================================================
 * macr_oracle.c -- CPU restatement of the MACR hot path.   TEST INFRASTRUCTURE.
 *
 * This file is the *checker*: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (macr_amd/) never
 * imports, links or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - evaluator half (top-K, masking, metrics): PINNED against golden vectors
 *     produced by running the reference's own evaluators here
 *     (tests/golden/make_golden.py: G5-G8) and against the reference's C++
 *     evaluator compiled from /root/reference into oracle/_ref.
 *   - adjacency / propagation: PINNED against G4 (reference get_adj_mat).
 *   - model-step half: losses, gradients, propagation and the test-time score
 *     formulas PINNED against fixture G10 (tests/golden/make_golden_model.py):
 *     the reference's own graph-building code (macr_mf/model.py loss builders,
 *     LightGCN._create_lightgcn_embed / create_bce_loss*) executed on injected
 *     tensors through a functional stand-in for the tensorflow module, with
 *     autograd gradients of that execution (tests/test_oracle_pinned.py).
 *     The optimizer -- tf.train.AdamOptimizer's update rule, epsilon placement,
 *     bias correction and its dense net effect on IndexedSlices -- stays PARITY
 *     UNPINNED: TensorFlow 1.14 cannot be run in this environment and the
 *     reference ships no golden values; orc_adam_dense follows SURVEY.md A.2
 *     (the rule as tf.train.AdamOptimizer's documentation states it, held to
 *     its float64 evaluation over gradients from 1e-12 to 1e-1 -- epsilon
 *     placement and bias correction included -- by tests/test_oracle_model.py
 *     ::test_oracle_adam_known_answers_from_1e_12_to_1e_1; that pins the
 *     restatement to the DOCUMENTED rule, not to TF 1.14's kernels).
 *
 * All tensors are fp32 row-major, indices int32, like the TF placeholders
 * (macr_mf/model.py:27-29).  Reductions over many elements accumulate in
 * double and round once to fp32: TF's own reduction order is not reproducible,
 * and the double sum is the value every fp32 ordering approximates.
 * Compile with -ffp-contract=off (oracle/Makefile) so that no multiply-add is
 * fused except the explicit fmaf() chains.
 * ==========================================================================*/
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_LOSS_NORMALBCE   0   /* macr_mf/model.py:277-287, LightGCN.py:415-429 */
#define ORC_LOSS_RUBIBCEBOTH 1   /* macr_mf/model.py:185-222, LightGCN.py:495-532 */
#define ORC_LOSS_RUBIBCE     2    /* --train rubibce  macr_mf/model.py:158-183: item branch only */

#define ORC_SCORE_NORMAL    0    /* batch_ratings      model.py:45,  LightGCN.py:166 */
#define ORC_SCORE_RUBI_BOTH 1    /* rubi_ratings_both  model.py:199, LightGCN.py:509 */
#define ORC_SCORE_RUBI      2    /* rubi_ratings       model.py:141  (batch_ratings - c) * sig_i                  */
#define ORC_SCORE_DIRECT_MINUS 3 /* direct_minus_ratings       model.py:142  batch_ratings - c * sig_i            */
#define ORC_SCORE_DIRECT_MINUS_BOTH 4 /* direct_minus_ratings_both :201  batch_ratings - c * sig_i * sig_u        */

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* d/dx of -log(sigmoid(x)+eps)      (tf.log grad = dy/x, SigmoidGrad = dy*y*(1-y)) */
static inline float dneglog_sig(float s, float eps) { return -((s * (1.0f - s)) / (s + eps)); }
/* d/dy of -log((1-sigmoid(y))+eps) */
static inline float dneglog_1msig(float s, float eps) { return (s * (1.0f - s)) / ((1.0f - s) + eps); }

/* ---------------------------------------------------------------------------
 * Row gather  (tf.nn.embedding_lookup, macr_mf/model.py:35-37; LightGCN.py:145-150)
 * -------------------------------------------------------------------------*/
void orc_gather_rows(const float *table, const int32_t *idx, int B, int d, float *out) {
    for (int r = 0; r < B; ++r)
        memcpy(out + (size_t)r * d, table + (size_t)idx[r] * d, sizeof(float) * d);
}

/* Scatter-add of per-pair gradient rows into a dense gradient: the net effect of
 * IndexedSlices de-duplication + sparse apply in TF 1.14. */
void orc_scatter_add_rows(float *dense, const int32_t *idx, int B, int d, const float *rows) {
    for (int r = 0; r < B; ++r) {
        float *dst = dense + (size_t)idx[r] * d;
        const float *src = rows + (size_t)r * d;
        for (int k = 0; k < d; ++k) dst[k] += src[k];
    }
}

/* ---------------------------------------------------------------------------
 * Pair loss + gradient on already gathered rows.
 *
 * kind = ORC_LOSS_NORMALBCE      macr_mf/model.py:277-287
 *   p = sum(eu*ei,1), n = sum(eu*ej,1)                                  :278-279
 *   mf = mean(-log(sig(p)+1e-9) - log(1-sig(n)+1e-9))                   :282
 * kind = ORC_LOSS_RUBIBCEBOTH    macr_mf/model.py:185-222
 *   si = ei@w, sj = ej@w, su = eu@w_u                                   :194-196
 *   pos = p(B,) * sig(si)(B,1) * sig(su)(B,1)  -> (B,B) broadcast        :204
 *   neg = n(B,) * sig(sj)(B,1) * sig(su)(B,1)  -> (B,B)                  :205
 *   L_ori  = mean_{r,c}(-log(sig(pos)+1e-10) - log(1-sig(neg)+1e-10))    :211
 *   L_item = mean(-log(sig(si)+1e-10) - log(1-sig(sj)+1e-10))            :213
 *   L_user = mean(-log(sig(su)+1e-10) - log(1-sig(su)+1e-10))            :215
 *   mf = L_ori + alpha*L_item + beta*L_user                              :217
 * kind = ORC_LOSS_RUBIBCE        macr_mf/model.py:158-183 -- the same graph without the user branch:
 *   pos = p(B,) * sig(si)(B,1), neg = n(B,) * sig(sj)(B,1) (again (B,B), :172-173), L_ori :174, L_item :176,
 *   mf = L_ori + alpha*L_item :178; w_user receives no gradient.  Restated as RUBIBCEBOTH with sig(su) := 1.
 * The l2 regulariser (:219-221) is handled by orc_l2_reg because LightGCN
 * regularises the *ego* rows, not the propagated ones (LightGCN.py:525-527).
 *
 * Outputs: mf_parts[4] = {mf, L_ori, L_item, L_user}; deu/dei/dej (B,d);
 * dw, dwu (d) are ACCUMULATED INTO (caller zeroes); fwd[5*B] = p,n,si,sj,su.
 * -------------------------------------------------------------------------*/
void orc_pair_loss_grad(int kind, int B, int d,
                        const float *eu, const float *ei, const float *ej,
                        const float *w, const float *wu, float alpha, float beta,
                        float *mf_parts, float *deu, float *dei, float *dej,
                        float *dw, float *dwu, float *fwd) {
    float *p = fwd, *n = fwd + B, *si = fwd + 2 * (size_t)B, *sj = fwd + 3 * (size_t)B,
          *su = fwd + 4 * (size_t)B;
    for (int r = 0; r < B; ++r) {
        double ap = 0, an = 0, asi = 0, asj = 0, asu = 0;
        const float *u = eu + (size_t)r * d, *a = ei + (size_t)r * d, *b = ej + (size_t)r * d;
        for (int k = 0; k < d; ++k) {
            ap += (double)(u[k] * a[k]);
            an += (double)(u[k] * b[k]);
            if (kind != ORC_LOSS_NORMALBCE) {
                asi += (double)(a[k] * w[k]);
                asj += (double)(b[k] * w[k]);
                if (kind == ORC_LOSS_RUBIBCEBOTH) asu += (double)(u[k] * wu[k]);
            }
        }
        p[r] = (float)ap; n[r] = (float)an;
        si[r] = (float)asi; sj[r] = (float)asj; su[r] = (float)asu;
    }
    float *dp = (float *)calloc(B, sizeof(float)), *dn = (float *)calloc(B, sizeof(float));
    float *dsi = (float *)calloc(B, sizeof(float)), *dsj = (float *)calloc(B, sizeof(float));
    float *dsu = (float *)calloc(B, sizeof(float));
    if (kind == ORC_LOSS_NORMALBCE) {
        const float eps = 1e-9f;
        double acc = 0;
        for (int r = 0; r < B; ++r) {
            float sp = sigmoidf_(p[r]), sn = sigmoidf_(n[r]);
            acc += (double)(-logf(sp + eps) + -logf((1.0f - sn) + eps));
            dp[r] = dneglog_sig(sp, eps) / (float)B;
            dn[r] = dneglog_1msig(sn, eps) / (float)B;
        }
        mf_parts[0] = (float)(acc / B); mf_parts[1] = mf_parts[0];
        mf_parts[2] = 0.f; mf_parts[3] = 0.f;
    } else {
        const float eps = 1e-10f;
        float *a = (float *)malloc(sizeof(float) * B), *b = (float *)malloc(sizeof(float) * B);
        float *ssi = (float *)malloc(sizeof(float) * B), *ssj = (float *)malloc(sizeof(float) * B);
        float *ssu = (float *)malloc(sizeof(float) * B);
        for (int r = 0; r < B; ++r) {
            ssi[r] = sigmoidf_(si[r]); ssj[r] = sigmoidf_(sj[r]);
            ssu[r] = kind == ORC_LOSS_RUBIBCEBOTH ? sigmoidf_(su[r]) : 1.0f;     /* RUBIBCE: no user factor */
            a[r] = ssi[r] * ssu[r];            /* row factor of `pos` (model.py:204) */
            b[r] = ssj[r] * ssu[r];            /* row factor of `neg` (model.py:205) */
        }
        double *da = (double *)calloc(B, sizeof(double)), *db = (double *)calloc(B, sizeof(double));
        double *dpd = (double *)calloc(B, sizeof(double)), *dnd = (double *)calloc(B, sizeof(double));
        double l_ori = 0;
        const double inv_b2 = 1.0 / ((double)B * (double)B);
        /* the (B,B) term: X[r,c] = p[c]*ssi[r]*ssu[r], Y[r,c] = n[c]*ssj[r]*ssu[r] */
#pragma omp parallel
        {
            double *dp_loc = (double *)calloc(B, sizeof(double));
            double *dn_loc = (double *)calloc(B, sizeof(double));
            double l_loc = 0;
#pragma omp for schedule(static)
            for (int r = 0; r < B; ++r) {
                double da_r = 0, db_r = 0;
                for (int c = 0; c < B; ++c) {
                    /* model.py:204-205 as written: pos_scores * sigmoid(item) * sigmoid(user), left to right (the
                     * gradient sums below go through a[r] = ssi*ssu: the same derivative) */
                    float x = (p[c] * ssi[r]) * ssu[r], y = (n[c] * ssj[r]) * ssu[r];
                    float sx = sigmoidf_(x), sy = sigmoidf_(y);
                    l_loc += (double)(-logf(sx + eps) + -logf((1.0f - sy) + eps));
                    float gx = dneglog_sig(sx, eps), gy = dneglog_1msig(sy, eps);
                    dp_loc[c] += (double)(gx * a[r]);
                    da_r += (double)(gx * p[c]);
                    dn_loc[c] += (double)(gy * b[r]);
                    db_r += (double)(gy * n[c]);
                }
                da[r] = da_r; db[r] = db_r;
            }
#pragma omp critical
            {
                l_ori += l_loc;
                for (int c = 0; c < B; ++c) { dpd[c] += dp_loc[c]; dnd[c] += dn_loc[c]; }
            }
            free(dp_loc); free(dn_loc);
        }
        double l_item = 0, l_user = 0;
        for (int r = 0; r < B; ++r) {
            l_item += (double)(-logf(ssi[r] + eps) + -logf((1.0f - ssj[r]) + eps));
            if (kind == ORC_LOSS_RUBIBCEBOTH) l_user += (double)(-logf(ssu[r] + eps) + -logf((1.0f - ssu[r]) + eps));
            float da_f = (float)(da[r] * inv_b2), db_f = (float)(db[r] * inv_b2);
            dp[r] = (float)(dpd[r] * inv_b2);
            dn[r] = (float)(dnd[r] * inv_b2);
            float dsig_i = ssi[r] * (1.0f - ssi[r]), dsig_j = ssj[r] * (1.0f - ssj[r]);
            float dsig_u = ssu[r] * (1.0f - ssu[r]);
            dsi[r] = da_f * dsig_i * ssu[r] + (alpha / (float)B) * dneglog_sig(ssi[r], eps);
            dsj[r] = db_f * dsig_j * ssu[r] + (alpha / (float)B) * dneglog_1msig(ssj[r], eps);
            dsu[r] = kind != ORC_LOSS_RUBIBCEBOTH ? 0.0f : (da_f * ssi[r] + db_f * ssj[r]) * dsig_u +
                     (beta / (float)B) * (dneglog_sig(ssu[r], eps) + dneglog_1msig(ssu[r], eps));
        }
        float Lo = (float)(l_ori * inv_b2), Li = (float)(l_item / B), Lu = (float)(l_user / B);
        mf_parts[1] = Lo; mf_parts[2] = Li; mf_parts[3] = Lu;
        mf_parts[0] = kind == ORC_LOSS_RUBIBCEBOTH ? Lo + alpha * Li + beta * Lu : Lo + alpha * Li;
        free(a); free(b); free(ssi); free(ssj); free(ssu);
        free(da); free(db); free(dpd); free(dnd);
    }
    /* back to the rows */
    double *dwd = (double *)calloc(d, sizeof(double)), *dwud = (double *)calloc(d, sizeof(double));
    for (int r = 0; r < B; ++r) {
        const float *u = eu + (size_t)r * d, *a = ei + (size_t)r * d, *b = ej + (size_t)r * d;
        float *gu = deu + (size_t)r * d, *ga = dei + (size_t)r * d, *gb = dej + (size_t)r * d;
        for (int k = 0; k < d; ++k) {
            float gu_k = dp[r] * a[k] + dn[r] * b[k];
            float ga_k = dp[r] * u[k];
            float gb_k = dn[r] * u[k];
            if (kind != ORC_LOSS_NORMALBCE) {
                if (kind == ORC_LOSS_RUBIBCEBOTH) gu_k += dsu[r] * wu[k];
                ga_k += dsi[r] * w[k];
                gb_k += dsj[r] * w[k];
                dwd[k] += (double)(a[k] * dsi[r]) + (double)(b[k] * dsj[r]);
                dwud[k] += (double)(u[k] * dsu[r]);
            }
            gu[k] = gu_k; ga[k] = ga_k; gb[k] = gb_k;
        }
    }
    if (kind != ORC_LOSS_NORMALBCE)
        for (int k = 0; k < d; ++k) { dw[k] += (float)dwd[k]; dwu[k] += (float)dwud[k]; }
    free(dwd); free(dwud);
    free(dp); free(dn); free(dsi); free(dsj); free(dsu);
}

/* ---------------------------------------------------------------------------
 * l2 regulariser on three gathered (ego) row blocks.
 *   regularizer = l2_loss(u)+l2_loss(i)+l2_loss(j)  (l2_loss = sum(x^2)/2)
 *   reg_loss = decay * regularizer / batch_size       macr_mf/model.py:219-221
 * batch_size is the CONFIGURED batch size (args.batch_size), not len(users).
 * Adds (decay/batch_size)*row to the three gradient blocks.
 * -------------------------------------------------------------------------*/
float orc_l2_reg(int B, int d, const float *eu, const float *ei, const float *ej,
                 float decay, int batch_size_cfg, float *deu, float *dei, float *dej) {
    double s = 0;
    const float coef = decay / (float)batch_size_cfg;
    for (size_t t = 0; t < (size_t)B * d; ++t) {
        s += (double)(eu[t] * eu[t]) + (double)(ei[t] * ei[t]) + (double)(ej[t] * ej[t]);
        if (deu) { deu[t] += coef * eu[t]; dei[t] += coef * ei[t]; dej[t] += coef * ej[t]; }
    }
    float regularizer = (float)(0.5 * s);
    regularizer = regularizer / (float)batch_size_cfg;
    return decay * regularizer;
}

/* ---------------------------------------------------------------------------
 * Adam exactly as tf.train.AdamOptimizer (TF 1.14) applies it to a variable
 * with the (de-duplicated, densified) gradient g -- SURVEY.md A.2:
 *   lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t);  m = b1*m+(1-b1)*g;
 *   v = b2*v+(1-b2)*g*g;  theta -= lr_t*m/(sqrt(v)+eps)
 * beta powers are fp32 state multiplied once per step (TF keeps them as fp32
 * variables): power[0]=beta1^t, power[1]=beta2^t on entry for step t.
 * -------------------------------------------------------------------------*/
float orc_adam_lr_t(float lr, const float *power) {
    return lr * sqrtf(1.0f - power[1]) / (1.0f - power[0]);
}

void orc_adam_dense(float *theta, float *m, float *v, const float *g, size_t n,
                    float lr_t, float b1, float b2, float eps) {
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < n; ++t) {
        float gt = g ? g[t] : 0.0f;
        float mt = m[t] * b1 + gt * (1.0f - b1);
        float vt = v[t] * b2 + (gt * gt) * (1.0f - b2);
        m[t] = mt; v[t] = vt;
        theta[t] = theta[t] - (lr_t * mt) / (sqrtf(vt) + eps);
    }
}

/* ---------------------------------------------------------------------------
 * One whole MF training step (macr_mf/train.py:487-496 -> model.py:74/:95).
 * P(n_users,d) Q(n_items,d) w(d) wu(d) and their Adam slots are updated in
 * place; power[2] is advanced.  losses = {loss, mf_loss, reg_loss}.
 * -------------------------------------------------------------------------*/
void orc_mf_train_step(int kind, int B, int d, int n_users, int n_items,
                       const int32_t *u, const int32_t *i, const int32_t *j,
                       float *P, float *Q, float *w, float *wu,
                       float *mP, float *vP, float *mQ, float *vQ,
                       float *mw, float *vw, float *mwu, float *vwu, float *power,
                       float lr, float b1, float b2, float eps,
                       float decay, float alpha, float beta, int batch_size_cfg,
                       float *losses) {
    size_t bd = (size_t)B * d;
    float *eu = (float *)malloc(bd * 4), *ei = (float *)malloc(bd * 4), *ej = (float *)malloc(bd * 4);
    float *deu = (float *)calloc(bd, 4), *dei = (float *)calloc(bd, 4), *dej = (float *)calloc(bd, 4);
    float *fwd = (float *)calloc((size_t)5 * B, 4);
#ifdef ORC_PERSISTENT_SCRATCH
    /* SPEED BUILD ONLY (oracle/Makefile `fast`, bench.py's tuned CPU port): the dense gradient tables are kept between calls
     * and are ALL ZERO between calls (the rows a batch touched are cleared again at the end of its step) instead of an 18 MB
     * calloc per step on the Gowalla shape.  Not re-entrant, not thread-safe, shared by models of equal size: never the checker. */
    static float *s_gP = NULL, *s_gQ = NULL;
    static size_t s_nP = 0, s_nQ = 0;
    if (s_nP != (size_t)n_users * d) { free(s_gP); s_nP = (size_t)n_users * d; s_gP = (float *)calloc(s_nP, 4); }
    if (s_nQ != (size_t)n_items * d) { free(s_gQ); s_nQ = (size_t)n_items * d; s_gQ = (float *)calloc(s_nQ, 4); }
    float *gP = s_gP, *gQ = s_gQ;
#else
    /* the checker: fresh zeroed gradient tables per call, freed on return -- re-entrant, no state between calls */
    float *gP = (float *)calloc((size_t)n_users * d, 4), *gQ = (float *)calloc((size_t)n_items * d, 4);
#endif
    float *gw = (float *)calloc(d, 4), *gwu = (float *)calloc(d, 4);
    float parts[4];
    orc_gather_rows(P, u, B, d, eu); orc_gather_rows(Q, i, B, d, ei); orc_gather_rows(Q, j, B, d, ej);
    orc_pair_loss_grad(kind, B, d, eu, ei, ej, w, wu, alpha, beta, parts, deu, dei, dej, gw, gwu, fwd);
    float reg = orc_l2_reg(B, d, eu, ei, ej, decay, batch_size_cfg, deu, dei, dej);
    orc_scatter_add_rows(gP, u, B, d, deu);
    orc_scatter_add_rows(gQ, i, B, d, dei);
    orc_scatter_add_rows(gQ, j, B, d, dej);
    float lr_t = orc_adam_lr_t(lr, power);
    orc_adam_dense(P, mP, vP, gP, (size_t)n_users * d, lr_t, b1, b2, eps);
    orc_adam_dense(Q, mQ, vQ, gQ, (size_t)n_items * d, lr_t, b1, b2, eps);
    if (kind != ORC_LOSS_NORMALBCE)               /* w only gets a gradient in the branch losses */
        orc_adam_dense(w, mw, vw, gw, d, lr_t, b1, b2, eps);
    if (kind == ORC_LOSS_RUBIBCEBOTH)             /* w_user: gradient None outside rubibceboth -> untouched */
        orc_adam_dense(wu, mwu, vwu, gwu, d, lr_t, b1, b2, eps);
    power[0] *= b1; power[1] *= b2;
    losses[1] = parts[0]; losses[2] = reg; losses[0] = parts[0] + reg;   /* model.py:73,:94 */
#ifdef ORC_PERSISTENT_SCRATCH
    for (int r = 0; r < B; ++r) {                  /* the touched rows are zero again */
        memset(gP + (size_t)u[r] * d, 0, (size_t)d * 4);
        memset(gQ + (size_t)i[r] * d, 0, (size_t)d * 4);
        memset(gQ + (size_t)j[r] * d, 0, (size_t)d * 4);
    }
#else
    free(gP); free(gQ);
#endif
    free(eu); free(ei); free(ej); free(deu); free(dei); free(dej); free(fwd);
    free(gw); free(gwu);
}

/* ---------------------------------------------------------------------------
 * LightGCN propagation:  Y = A_hat @ X  with A_hat in CSR
 * (tf.sparse_tensor_dense_matmul, LightGCN.py:301; the 100 row folds of
 * _split_A_hat :257-269 concatenate to exactly this product).  fp32,
 * accumulation in stored (column-sorted) order like TF's CPU kernel.
 * -------------------------------------------------------------------------*/
void orc_spmm_csr(int n_rows, int d, const int32_t *indptr, const int32_t *indices,
                  const float *data, const float *X, float *Y) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int r = 0; r < n_rows; ++r) {
        float *y = Y + (size_t)r * d;
        for (int k = 0; k < d; ++k) y[k] = 0.0f;
        for (int e = indptr[r]; e < indptr[r + 1]; ++e) {
            const float a = data[e];
            const float *x = X + (size_t)indices[e] * d;
            for (int k = 0; k < d; ++k) y[k] += a * x[k];
        }
    }
}

/* E = mean(E0, A E0, A^2 E0, ...) over n_layers+1 terms  (LightGCN.py:288-309).
 * work must hold 2*N*d floats. */
void orc_lgcn_propagate(int N, int d, int n_layers, const int32_t *indptr, const int32_t *indices,
                        const float *data, const float *E0, float *E, float *work) {
    size_t nd = (size_t)N * d;
    float *cur = work, *nxt = work + nd;
    memcpy(cur, E0, nd * 4);
    memcpy(E, E0, nd * 4);
    for (int l = 0; l < n_layers; ++l) {
        orc_spmm_csr(N, d, indptr, indices, data, cur, nxt);
        for (size_t t = 0; t < nd; ++t) E[t] += nxt[t];
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    const float inv = 1.0f / (float)(n_layers + 1);     /* reduce_mean over the stack */
    for (size_t t = 0; t < nd; ++t) E[t] = E[t] * inv;
}

/* Backward of the above wrt E0 given dE:  dE0 = (dE + A^T dE + (A^T)^2 dE + ...)/(L+1) -- the same propagation with the
 * TRANSPOSED operator.  The caller passes A^T in CSR; for the symmetric `pre` / `plain` matrices of the README's commands
 * that is A itself.  work: 2*N*d floats. */
void orc_lgcn_propagate_bwd(int N, int d, int n_layers, const int32_t *indptr, const int32_t *indices,
                            const float *data, const float *dE, float *dE0, float *work) {
    orc_lgcn_propagate(N, d, n_layers, indptr, indices, data, dE, dE0, work);
}

/* One whole LightGCN training step (LightGCN.py:598-607 -> :186/:201).
 * T = [P;Q] is the (N,d) ego table, N = n_users+n_items. */
/* indptr_t / indices_t / adj_t: the TRANSPOSED adjacency in CSR, used by the backward pass (tf.gradients of
 * tf.sparse_tensor_dense_matmul, LightGCN.py:301): --adj_type norm / gcmc / mean are row-normalised, D^-1 A, and not symmetric
 * (utility/load_data.py:95-164, LightGCN.py:667-678). */
void orc_lgcn_train_step_t(int kind, int B, int d, int n_users, int n_items, int n_layers,
                         const int32_t *indptr, const int32_t *indices, const float *adj,
                         const int32_t *indptr_t, const int32_t *indices_t, const float *adj_t,
                         const int32_t *u, const int32_t *i, const int32_t *j,
                         float *T, float *w, float *wu, float *mT, float *vT,
                         float *mw, float *vw, float *mwu, float *vwu, float *power,
                         float lr, float b1, float b2, float eps,
                         float decay, float alpha, float beta, int batch_size_cfg, float *losses) {
    int N = n_users + n_items;
    size_t nd = (size_t)N * d, bd = (size_t)B * d;
    float *E = (float *)malloc(nd * 4), *work = (float *)malloc(2 * nd * 4);
    float *dE = (float *)calloc(nd, 4), *G = (float *)malloc(nd * 4);
    int32_t *ii = (int32_t *)malloc(4 * B), *jj = (int32_t *)malloc(4 * B);
    for (int r = 0; r < B; ++r) { ii[r] = i[r] + n_users; jj[r] = j[r] + n_users; }
    orc_lgcn_propagate(N, d, n_layers, indptr, indices, adj, T, E, work);
    float *eu = (float *)malloc(bd * 4), *ei = (float *)malloc(bd * 4), *ej = (float *)malloc(bd * 4);
    float *deu = (float *)calloc(bd, 4), *dei = (float *)calloc(bd, 4), *dej = (float *)calloc(bd, 4);
    float *fwd = (float *)calloc((size_t)5 * B, 4);
    float *gw = (float *)calloc(d, 4), *gwu = (float *)calloc(d, 4);
    float parts[4];
    orc_gather_rows(E, u, B, d, eu); orc_gather_rows(E, ii, B, d, ei); orc_gather_rows(E, jj, B, d, ej);
    orc_pair_loss_grad(kind, B, d, eu, ei, ej, w, wu, alpha, beta, parts, deu, dei, dej, gw, gwu, fwd);
    orc_scatter_add_rows(dE, u, B, d, deu);
    orc_scatter_add_rows(dE, ii, B, d, dei);
    orc_scatter_add_rows(dE, jj, B, d, dej);
    orc_lgcn_propagate_bwd(N, d, n_layers, indptr_t, indices_t, adj_t, dE, G, work);
    /* regulariser acts on the ego rows (LightGCN.py:525-528) */
    orc_gather_rows(T, u, B, d, eu); orc_gather_rows(T, ii, B, d, ei); orc_gather_rows(T, jj, B, d, ej);
    memset(deu, 0, bd * 4); memset(dei, 0, bd * 4); memset(dej, 0, bd * 4);
    float reg = orc_l2_reg(B, d, eu, ei, ej, decay, batch_size_cfg, deu, dei, dej);
    orc_scatter_add_rows(G, u, B, d, deu);
    orc_scatter_add_rows(G, ii, B, d, dei);
    orc_scatter_add_rows(G, jj, B, d, dej);
    float lr_t = orc_adam_lr_t(lr, power);
    orc_adam_dense(T, mT, vT, G, nd, lr_t, b1, b2, eps);
    if (kind == ORC_LOSS_RUBIBCEBOTH) {
        orc_adam_dense(w, mw, vw, gw, d, lr_t, b1, b2, eps);
        orc_adam_dense(wu, mwu, vwu, gwu, d, lr_t, b1, b2, eps);
    }
    power[0] *= b1; power[1] *= b2;
    losses[1] = parts[0]; losses[2] = reg; losses[0] = parts[0] + reg;
    free(E); free(work); free(dE); free(G); free(ii); free(jj);
    free(eu); free(ei); free(ej); free(deu); free(dei); free(dej); free(fwd); free(gw); free(gwu);
}

/* the same for a symmetric adjacency (A^T = A): what the README's commands run (--adj_type pre) */
void orc_lgcn_train_step(int kind, int B, int d, int n_users, int n_items, int n_layers,
                         const int32_t *indptr, const int32_t *indices, const float *adj,
                         const int32_t *u, const int32_t *i, const int32_t *j,
                         float *T, float *w, float *wu, float *mT, float *vT,
                         float *mw, float *vw, float *mwu, float *vwu, float *power,
                         float lr, float b1, float b2, float eps,
                         float decay, float alpha, float beta, int batch_size_cfg, float *losses) {
    orc_lgcn_train_step_t(kind, B, d, n_users, n_items, n_layers, indptr, indices, adj, indptr, indices, adj, u, i, j, T, w, wu,
                          mT, vT, mw, vw, mwu, vwu, power, lr, b1, b2, eps, decay, alpha, beta, batch_size_cfg, losses);
}

/* ---------------------------------------------------------------------------
 * Branch sigmoids used at test time:  out[r] = sigmoid(rows[r] . w)
 * (tf.nn.sigmoid(tf.matmul(e, w)), macr_mf/model.py:194-196,:199)
 * -------------------------------------------------------------------------*/
void orc_branch_sigmoid(const float *rows, int n, int d, const float *w, float *out) {
    for (int r = 0; r < n; ++r) {
        double s = 0;
        for (int k = 0; k < d; ++k) s += (double)(rows[(size_t)r * d + k] * w[k]);
        out[r] = sigmoidf_((float)s);
    }
}

/* Test-time score of one (user, item) from its dot product `acc`; each operation rounds on its own (no fused
 * multiply-add), in the order the reference's expression evaluates:
 *   NORMAL             acc                                                    model.py:45
 *   RUBI_BOTH          ((acc - c) * sig_i) * sig_u                            model.py:199
 *   RUBI               (acc - c) * sig_i                                      model.py:141
 *   DIRECT_MINUS       acc - (c * sig_i)                                      model.py:142
 *   DIRECT_MINUS_BOTH  acc - ((c * sig_i) * sig_u)                            model.py:201 */
static inline float orc_score_epilogue(int kind, float acc, float c, float sgi, float su) {
    volatile float t;
    switch (kind) {
        case ORC_SCORE_RUBI_BOTH: t = acc - c; t = t * sgi; t = t * su; return t;
        case ORC_SCORE_RUBI: t = acc - c; t = t * sgi; return t;
        case ORC_SCORE_DIRECT_MINUS: t = c * sgi; t = acc - t; return t;
        case ORC_SCORE_DIRECT_MINUS_BOTH: t = c * sgi; t = t * su; t = acc - t; return t;
        default: return acc;
    }
}

/* Dense (U,N) scores: the literal sess.run(model.<ratings>, {users, pos_items: all items}) (macr_mf/train.py:224-251). */
void orc_score_matrix(int kind, int U, int N, int d, const float *Urows, const float *Irows,
                      const float *sig_u, const float *sig_i, float c, float *out) {
#pragma omp parallel for schedule(static)
    for (int u = 0; u < U; ++u)
        for (int it = 0; it < N; ++it) {
            const float *ur = Urows + (size_t)u * d, *ir = Irows + (size_t)it * d;
            float acc = 0.0f;
            for (int k = 0; k < d; ++k) acc = fmaf(ur[k], ir[k], acc);
            out[(size_t)u * N + it] = orc_score_epilogue(kind, acc, c, sig_i ? sig_i[it] : 1.0f, sig_u ? sig_u[u] : 1.0f);
        }
}

/* ---------------------------------------------------------------------------
 * Full-catalogue scoring + train-item masking + top-K.
 *   NORMAL    : S[u,i] = U[u].I[i]                              model.py:45
 *   RUBI_BOTH : S[u,i] = (U[u].I[i] - c) * sig_i[i] * sig_u[u]  model.py:199
 * The dot product is an fmaf chain in k order from 0 -- the arithmetic of the
 * gfx950 fp32 MFMA, so the HIP path is expected to agree bit for bit.
 * Masking: items listed in the user's (sorted) mask row are removed from the
 * candidates (macr_mf/train.py:132-133; = set to -inf in
 * macr_lightgcn/utility/batch_test.py:124-129).
 * Ranking: descending score, exact ties -> lower item id first.  This is what
 * heapq.nlargest over the ascending candidate list does (train.py:89-104);
 * std::partial_sort_copy (tools.h:13-22) leaves tie order unspecified.
 * item_offset is added to local item ids (item-sharded evaluation).
 * out_idx/out_val are (U,K); out_cnt[u] = min(K, #candidates).  Unused slots:
 * idx -1 / val -inf, or when fill_masked != 0 the user's masked items in
 * ascending id order with val -inf (what a -inf masked matrix would rank next).
 * -------------------------------------------------------------------------*/
void orc_score_topk(int kind, int U, int N, int d, const float *Urows, const float *Irows,
                    const float *sig_u, const float *sig_i, float c,
                    const int32_t *mask_ptr, const int32_t *mask_idx, int item_offset,
                    int K, int fill_masked, float *out_val, int32_t *out_idx, int32_t *out_cnt) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int u = 0; u < U; ++u) {
        float *bv = out_val + (size_t)u * K;
        int32_t *bi = out_idx + (size_t)u * K;
        int cnt = 0;
        int mp = mask_ptr ? mask_ptr[u] : 0, me = mask_ptr ? mask_ptr[u + 1] : 0;
        while (mp < me && mask_idx[mp] < item_offset) ++mp;
        const float *ur = Urows + (size_t)u * d;
        for (int it = 0; it < N; ++it) {
            int gid = it + item_offset;
            if (mp < me && mask_idx[mp] == gid) { ++mp; continue; }
            const float *ir = Irows + (size_t)it * d;
            float acc = 0.0f;
            for (int k = 0; k < d; ++k) acc = fmaf(ur[k], ir[k], acc);
            float s = acc;
            s = orc_score_epilogue(kind, acc, c, sig_i ? sig_i[it] : 1.0f, sig_u ? sig_u[u] : 1.0f);
            /* insert: strictly greater moves ahead of equal (earlier id stays first) */
            if (cnt < K || s > bv[cnt - 1]) {
                int pos = cnt < K ? cnt : K - 1;
                while (pos > 0 && s > bv[pos - 1]) { bv[pos] = bv[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bv[pos] = s; bi[pos] = gid;
                if (cnt < K) ++cnt;
            }
        }
        out_cnt[u] = cnt;
        int slot = cnt;
        if (fill_masked && mask_ptr)
            for (int e = mask_ptr[u]; e < mask_ptr[u + 1] && slot < K; ++e) {
                int gid = mask_idx[e];
                if (gid < item_offset || gid >= item_offset + N) continue;
                bv[slot] = -INFINITY; bi[slot] = gid; ++slot;
            }
        for (; slot < K; ++slot) { bv[slot] = -INFINITY; bi[slot] = -1; }
    }
}

/* Top-K straight from a score matrix (U,N): the job of c_top_k_array_index
 * (macr_lightgcn/evaluator/cpp/include/tools.h:24) with the tie rule fixed to
 * "lower id first".  An optional mask CSR removes candidates exactly as in
 * orc_score_topk; without a mask every column (including -inf ones, as
 * produced by batch_test.py:129) is a candidate. */
void orc_topk_scores(int U, int N, const float *scores, const int32_t *mask_ptr,
                     const int32_t *mask_idx, int K, float *out_val, int32_t *out_idx,
                     int32_t *out_cnt) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int u = 0; u < U; ++u) {
        float *bv = out_val + (size_t)u * K;
        int32_t *bi = out_idx + (size_t)u * K;
        const float *row = scores + (size_t)u * N;
        int cnt = 0;
        int mp = mask_ptr ? mask_ptr[u] : 0, me = mask_ptr ? mask_ptr[u + 1] : 0;
        for (int it = 0; it < N; ++it) {
            if (mp < me && mask_idx[mp] == it) { ++mp; continue; }
            float s = row[it];
            if (cnt < K || s > bv[cnt - 1]) {
                int pos = cnt < K ? cnt : K - 1;
                while (pos > 0 && s > bv[pos - 1]) { bv[pos] = bv[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bv[pos] = s; bi[pos] = it;
                if (cnt < K) ++cnt;
            }
        }
        out_cnt[u] = cnt;
        for (int k = cnt; k < K; ++k) { bv[k] = -INFINITY; bi[k] = -1; }
    }
}

/* Merge W per-shard top-K lists (W,U,K) into one (U,K): same order as above
 * (score desc, id asc).  Entries with idx < 0 are empty; entries with
 * val == -inf and idx >= 0 are masked fill and are ordered after every real
 * candidate by id.  Exact: global top-K is a subset of the union. */
void orc_topk_merge(int W, int U, int K, const float *vals, const int32_t *idxs,
                    float *out_val, int32_t *out_idx, int32_t *out_cnt) {
    for (int u = 0; u < U; ++u) {
        float *bv = out_val + (size_t)u * K;
        int32_t *bi = out_idx + (size_t)u * K;
        int cnt = 0, real = 0;
        for (int s = 0; s < W; ++s)
            for (int k = 0; k < K; ++k) {
                size_t off = ((size_t)s * U + u) * K + k;
                float v = vals[off]; int32_t id = idxs[off];
                if (id < 0) continue;
                int pos = cnt < K ? cnt : K;
                while (pos > 0 && (v > bv[pos - 1] || (v == bv[pos - 1] && id < bi[pos - 1]))) --pos;
                if (pos >= K) continue;
                int last = cnt < K ? cnt : K - 1;
                for (int t = last; t > pos; --t) { bv[t] = bv[t - 1]; bi[t] = bi[t - 1]; }
                bv[pos] = v; bi[pos] = id;
                if (cnt < K) ++cnt;
            }
        for (int k = 0; k < cnt; ++k) if (bv[k] > -INFINITY) ++real;
        out_cnt[u] = real;
        for (int k = cnt; k < K; ++k) { bv[k] = -INFINITY; bi[k] = -1; }
    }
}

static int in_sorted(const int32_t *a, int n, int32_t x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == x;
}

/* ---------------------------------------------------------------------------
 * Fold-out metrics, restating macr_lightgcn/evaluator/cpp/include/
 * evaluate_foldout.h:16-195: for every user 5 metrics x K prefix values, laid
 * out [precision | recall | ap | ndcg | mrr], float accumulators with the
 * double-precision sub-expressions the C++ has (1.0*hits/(i+1), 1.0/log2(i+2)).
 * gt rows (ground truth) must be sorted ascending.  rankings may contain -1.
 * -------------------------------------------------------------------------*/
void orc_metrics_foldout(int U, int K, const int32_t *rankings, const int32_t *gt_ptr,
                         const int32_t *gt_idx, float *results) {
    for (int u = 0; u < U; ++u) {
        const int32_t *rank = rankings + (size_t)u * K;
        const int32_t *truth = gt_idx + gt_ptr[u];
        int truth_len = gt_ptr[u + 1] - gt_ptr[u];
        float *res = results + (size_t)u * 5 * K;
        int hits = 0; float sum_pre = 0, dcg = 0, idcg = 0, rr = 0; int found = 0;
        for (int i = 0; i < K; ++i) {
            int hit = rank[i] >= 0 && in_sorted(truth, truth_len, rank[i]);
            if (hit) {
                hits += 1;
                float pre = (float)(1.0 * hits / (i + 1));
                sum_pre += pre;
                dcg = (float)(dcg + 1.0 / log2((double)(i + 2)));
                if (!found) { rr = (float)(1.0 / (i + 1)); found = 1; }
            }
            if (i < truth_len) idcg = (float)(idcg + 1.0 / log2((double)(i + 2)));
            res[0 * K + i] = (float)(1.0 * hits / (i + 1));
            res[1 * K + i] = (float)(1.0 * hits / truth_len);
            res[2 * K + i] = sum_pre / truth_len;
            res[3 * K + i] = dcg / idcg;
            res[4 * K + i] = rr;
        }
    }
}

/* ---------------------------------------------------------------------------
 * MF metrics, restating macr_mf/train.py:32-117 in float64 like NumPy:
 * for each user and each K in Ks: precision, recall, ndcg, hit_ratio from the
 * hit flags r of the top-max(Ks) list.  cnt[u] = length of that list (it is
 * shorter than K when the user has fewer candidates; np.mean(r[:K]) then
 * averages over the shorter list).  out is (U, 4, nK) doubles.
 * -------------------------------------------------------------------------*/
void orc_metrics_mf(int U, int Kmax, const int32_t *rankings, const int32_t *cnt,
                    const int32_t *gt_ptr, const int32_t *gt_idx,
                    const int32_t *Ks, int nK, double *out) {
    for (int u = 0; u < U; ++u) {
        const int32_t *rank = rankings + (size_t)u * Kmax;
        const int32_t *truth = gt_idx + gt_ptr[u];
        int truth_len = gt_ptr[u + 1] - gt_ptr[u];
        int len = cnt ? cnt[u] : Kmax;
        for (int q = 0; q < nK; ++q) {
            int K = Ks[q];
            int m = len < K ? len : K;
            double hits = 0, dcg = 0, dcg_max = 0;
            for (int i = 0; i < m; ++i)
                if (rank[i] >= 0 && in_sorted(truth, truth_len, rank[i])) {
                    hits += 1.0; dcg += 1.0 / log2((double)(i + 2));
                }
            int lim = truth_len < K ? truth_len : K;
            for (int i = 0; i < lim; ++i) dcg_max += 1.0 / log2((double)(i + 2));
            double *o = out + ((size_t)u * 4) * nK;
            o[0 * nK + q] = m > 0 ? hits / m : NAN;             /* precision_at_k :32-43 */
            o[1 * nK + q] = hits / truth_len;                   /* recall_at_k    :77-79 */
            o[2 * nK + q] = dcg_max != 0 ? dcg / dcg_max : 0.0; /* ndcg_at_k      :63-74 */
            o[3 * nK + q] = hits > 0 ? 1.0 : 0.0;               /* hit_at_k       :82-87 */
        }
    }
}

/* ---------------------------------------------------------------------------
 * Device sampler: the CHECKER of macr_amd/csrc/sample_kernels.hip.
 * The law is the reference's -- macr_mf/load_data.py:543-566 (Data.sample: B distinct users by rd.sample, or
 * rd.choice when B exceeds the user count; a uniform positive of the user's train list, item 0 for an empty list
 * :551-552; a uniform negative outside that list by rejection :554-558), macr_lightgcn/utility/load_data.py:174-212
 * (sample: users from exist_users) and :214-254 (sample_test: positives from the test list, negatives outside test
 * AND train lists) -- the STREAM is the kernel's own counter-based generator, restated here operation by operation
 * (all integer arithmetic: results are compared bit for bit):
 *   key     = mix64(seed * 0x9e3779b97f4a7c15 + step)                      splitmix64 finaliser
 *   draw(n) = high 32 bits of mix64(key ^ (triple << 32 | n))
 *   users   : B <= n_pool: image of `triple` under a 4-round Feistel permutation of [0, n_pool) (cycle walking
 *             over the next even power of two) -- distinct users; else below(draw(0), n_pool)
 *   pos     = list[below(draw(1), len)], 0 for an empty list
 *   neg     = first of below(draw(2)), below(draw(3)), ... (at most 4096 tries) outside the exclusion list
 *   below(r, range) = (r * range) >> 32
 * -------------------------------------------------------------------------*/
static inline uint64_t orc_mix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
static inline uint32_t orc_draw(uint64_t key, uint32_t t, uint32_t n) {
    return (uint32_t)(orc_mix64(key ^ ((uint64_t)t << 32 | n)) >> 32);
}
static inline uint32_t orc_below(uint32_t r, uint32_t range) { return (uint32_t)(((uint64_t)r * range) >> 32); }
static uint32_t orc_feistel_perm(uint32_t x, uint32_t n, int bits, uint64_t key) {
    const int hb = (bits + 1) / 2;
    const uint32_t mask = (1u << hb) - 1u;
    do {
        uint32_t l = x >> hb, r = x & mask;
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = (uint32_t)(orc_mix64(key + 0x1234567ull * (uint64_t)(round + 1) + r) >> 17) & mask;
            const uint32_t nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = (l << hb) | r;
    } while (x >= n);
    return x;
}
/* out: (3, B) int32 = users, positives, negatives of batch `step`.  pool / excl_ptr / excl_idx may be NULL. */
void orc_sample_triples(uint64_t seed, uint64_t step, int B, int n_items, const int32_t *pool, int n_pool,
                        const int32_t *train_ptr, const int32_t *train_idx,
                        const int32_t *excl_ptr, const int32_t *excl_idx, int32_t *out) {
    int bits = 1;
    while ((1u << bits) < (unsigned)n_pool) ++bits;
    if (bits & 1) ++bits;
    const uint64_t key = orc_mix64(seed * 0x9e3779b97f4a7c15ull + step);
    for (int t = 0; t < B; ++t) {
        uint32_t slot;
        if (B <= n_pool) slot = orc_feistel_perm((uint32_t)t, (uint32_t)n_pool, bits, key);
        else slot = orc_below(orc_draw(key, (uint32_t)t, 0), (uint32_t)n_pool);
        const int user = pool ? pool[slot] : (int)slot;
        const int beg = train_ptr[user], len = train_ptr[user + 1] - beg;
        const int pos = len > 0 ? train_idx[beg + (int)orc_below(orc_draw(key, (uint32_t)t, 1), (uint32_t)len)] : 0;
        const int32_t *xi = excl_ptr ? excl_idx : train_idx;
        const int xbeg = excl_ptr ? excl_ptr[user] : beg, xlen = excl_ptr ? excl_ptr[user + 1] - xbeg : len;
        int neg = 0;
        for (uint32_t n = 2; n < 2 + 4096; ++n) {
            neg = (int)orc_below(orc_draw(key, (uint32_t)t, n), (uint32_t)n_items);
            if (!in_sorted(xi + xbeg, xlen, neg)) break;
        }
        out[t] = user; out[B + t] = pos; out[2 * (size_t)B + t] = neg;
    }
}
