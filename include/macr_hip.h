/* ============================================================================
 * macr_hip.h -- C ABI of the MI355X (gfx950) hot path of MACR.
 *
 * libmacr_hip.so (built from macr_amd/csrc/ by hipcc) exports exactly the
 * symbols declared here.  Conventions, for every entry point:
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - every pointer marked (dev) is device memory owned by the CALLER
 *     (PyTorch tensors on the host side); the library never allocates, frees
 *     or retains device memory and keeps no global state;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it and nothing synchronises the device;
 *   - return value: MACR_OK (0) or a negative MACR_E_* code; on error nothing
 *     has been enqueued and macr_last_error() describes the problem;
 *   - fp32 row-major tensors, int32 indices (the reference's TF placeholders,
 *     macr_mf/model.py:27-29);
 *   - thread-compatible: concurrent calls are fine as long as they do not
 *     share output or workspace buffers.
 *
 * Each entry point names the reference code it replaces (file:line relative to
 * the weitianxin/MACR checkout).  The reference has one real C ABI on this
 * path -- c_top_k_array_index / evaluate_foldout, declared in
 * macr_lightgcn/evaluator/cpp/apt_evaluate_foldout.pyx:11-19 -- whose
 * replacements are macr_topk_scores and macr_metrics_foldout; everything else
 * replaces a `sess.run` of TF-1.14 stock ops.
 * ==========================================================================*/
#ifndef MACR_HIP_H
#define MACR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MACR_OK              0
#define MACR_E_INVALID      -1   /* bad argument (null pointer, negative size, ...)            */
#define MACR_E_UNSUPPORTED  -2   /* valid request this build has no kernel for (dim, K)         */
#define MACR_E_WORKSPACE    -3   /* workspace too small                                         */
#define MACR_E_LAUNCH       -4   /* hipLaunch / runtime failure                                 */

#define MACR_ABI_VERSION     15

/* loss kinds */
#define MACR_LOSS_NORMALBCE   0  /* --train normalbce   macr_mf/model.py:277-287 ; --loss bce     LightGCN.py:415-429 */
#define MACR_LOSS_RUBIBCEBOTH 1  /* --train rubibceboth macr_mf/model.py:185-222 ; --loss bceboth LightGCN.py:495-532 */
#define MACR_LOSS_RUBIBCE     2  /* --train rubibce     macr_mf/model.py:158-183 : item branch only (MF only);
                                    the rubibceboth graph with sigmoid(e_u.w_user) := 1, w_user untouched        */

/* score kinds */
#define MACR_SCORE_NORMAL    0   /* batch_ratings      macr_mf/model.py:45  ; LightGCN.py:166 */
#define MACR_SCORE_RUBI_BOTH 1   /* rubi_ratings_both  macr_mf/model.py:199 ; LightGCN.py:509   ((y - c) * sig_i) * sig_u */
#define MACR_SCORE_RUBI      2   /* rubi_ratings       macr_mf/model.py:141                      (y - c) * sig_i          */
#define MACR_SCORE_DIRECT_MINUS      3  /* direct_minus_ratings      model.py:142               y - (c * sig_i)           */
#define MACR_SCORE_DIRECT_MINUS_BOTH 4  /* direct_minus_ratings_both model.py:201 ; LightGCN.py:510  y - ((c * sig_i) * sig_u) */

/* largest K the top-K kernels are built for (the reference uses 20; parser default max 30) */
#define MACR_MAX_TOPK 128          /* macr_score_topk, macr_topk_merge, macr_topk_scores, the metrics (--Ks of the reference: any list) */
#define MACR_MAX_TOPK_FUSED 32     /* up to here macr_score_topk is the fused ranking (thresholds, candidate lists, seeds); above,
                                      blocks of dense score rows + streaming selection -- same results, no seeds; the c sweep
                                      (macr_score_topk_sweep) stays at K <= 32 */
/* macr_topk_scores itself takes ANY K, like c_top_k_array_index of the reference (tools.h:13-22): above MACR_MAX_TOPK it
 * ranks in rounds of MACR_MAX_TOPK positions and needs macr_topk_scores_workspace_bytes(rows, K) of scratch */
#define MACR_SEED_WIDTH 32       /* threshold seeds per query of macr_score_topk (seed_idx / seed_out) */

int         macr_abi_version(void);
const char *macr_last_error(void);          /* thread-local, valid until the next call     */
const char *macr_build_info(void);          /* "gfx950 hipcc <ver> ..."                    */

/* Optional per-kernel timing for benchmarks (per host thread).  Between
 * macr_timing_begin(stream) and macr_timing_end(), every kernel the library
 * launches on that stream is followed by a hipEvent; macr_timing_end waits for
 * the last one and returns how many (name, milliseconds) pairs it wrote
 * (names: max_n x 32 chars).  Not capturable into a hipGraph while active. */
int macr_timing_begin(void *stream);
int macr_timing_end(int max_n, char *names, float *ms);

/* Hyper-parameters of one training step (all by value, host side). */
typedef struct macr_hyper {
    float lr;              /* --lr                                                   */
    float beta1, beta2;    /* Adam 0.9 / 0.999  (tf.train.AdamOptimizer defaults)    */
    float adam_eps;        /* 1e-8                                                   */
    float decay;           /* --regs   (macr_mf/model.py:19 ; LightGCN.py:50)        */
    float alpha, beta;     /* --alpha --beta (item / user branch weights)            */
    int32_t batch_size_cfg;/* args.batch_size: divisor of the regulariser (:220)     */
} macr_hyper;

/* ---------------------------------------------------------------------------
 * Training step, matrix factorisation.
 * Replaces  sess.run([opt_X, loss_X, mf_loss_X, reg_loss_X], feed{users,pos,neg})
 * (macr_mf/train.py:487-496) i.e. gathers (model.py:35-37), loss (:185-222 or
 * :277-287), gradients, IndexedSlices de-duplication and tf.train.AdamOptimizer
 * (:74/:95) -- dense Adam semantics, see SURVEY.md A.2.
 *
 *   u,i,j      (dev) int32[B]     sampled users / positive / negative items
 *   P,Q        (dev) fp32[n_users*d], fp32[n_items*d]  updated in place
 *   w,wu       (dev) fp32[d]      item / user branch vectors (rubibceboth only)
 *   m*,v*      (dev) Adam slots, same shapes, updated in place
 *   gP,gQ      (dev) fp32 same shape as P,Q: dense gradient scratch that MUST be
 *              all-zero on entry and is all-zero again on return (rows touched
 *              by the batch are consumed and re-zeroed by the Adam kernel)
 *   touchedP,Q (dev) int32[n_users], int32[n_items]: same zero-in / zero-out rule
 *   adam_pow   (dev) fp32[2] = {beta1^t, beta2^t} for the step about to run
 *              (initialise to {beta1, beta2}); advanced on device
 *   losses     (dev) fp32[3] = {loss, mf_loss, reg_loss} of this step
 *   workspace  (dev) >= macr_mf_train_workspace_bytes(B, d) bytes, 256-B aligned
 *
 * d must be 32, 64, 128 or 256.  B >= 1.  No host synchronisation; safe to
 * capture into a hipGraph.
 *
 * flags = 0: the call is one complete step (P, Q, w, wu and the slots are up to
 * date in stream order when it returns).
 * Deferred mode (rubibceboth only): the step's pass over ALL rows of P and Q
 * (tf.train.AdamOptimizer moves every row every step) is bound by HBM, the
 * (B,B) loss kernel by the VALU; neither depends on the other across a step
 * boundary, so consecutive steps can overlap them:
 *   MACR_STEP_DEFER    leave this step's dense Adam pass pending.  On return the
 *                      losses are final, the gradient sums are in gP/gQ (flags
 *                      in touchedP/Q) and lr_t / the branch-vector partials are
 *                      in the workspace; P, Q, w, wu and the slots still hold
 *                      the previous values.
 *   MACR_STEP_PENDING  the previous call on these buffers used MACR_STEP_DEFER:
 *                      this call first completes that update for the rows its
 *                      own batch reads (and for w, wu), then runs the rest of the
 *                      pending pass as extra blocks of its (B,B) launch.  Needs
 *                      the same B, d, gP/gQ/touched buffers and the same,
 *                      unmodified workspace as the deferring call.
 * macr_mf_train_flush completes a pending pass on its own (before evaluation,
 * checkpointing, a change of B, or the end of training).  The arithmetic of every
 * row is identical in both modes; only the order of independent rows changes.
 * -------------------------------------------------------------------------*/
#define MACR_STEP_DEFER   1
#define MACR_STEP_PENDING 2
#define MACR_STEP_LOSS_ONLY 4    /* macr_lgcn_train_step only: compute the losses, update nothing (see there) */
#define MACR_STEP_DENSE_LAYERS 8 /* macr_lgcn_train_step only: every propagation layer dense (default: the last forward layer
                                    computes the batch's rows only, the first backward layer gathers from them only) */

size_t macr_mf_train_workspace_bytes(int B, int d);

int macr_mf_train_step(int loss_kind, int B, int d, int n_users, int n_items,
                       const int32_t *u, const int32_t *i, const int32_t *j,
                       float *P, float *Q, float *w, float *wu,
                       float *mP, float *vP, float *mQ, float *vQ,
                       float *mw, float *vw, float *mwu, float *vwu,
                       float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                       float *adam_pow, const macr_hyper *hp,
                       float *losses, int flags, void *workspace, size_t workspace_bytes, void *stream);

int macr_mf_train_flush(int loss_kind, int B, int d, int n_users, int n_items,
                        float *P, float *Q, float *w, float *wu,
                        float *mP, float *vP, float *mQ, float *vQ,
                        float *mw, float *vw, float *mwu, float *vwu,
                        float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                        const macr_hyper *hp, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Row-sharded training: the device half of one step when P, Q, their Adam slots and
 * gradient scratch are range-sharded over W ranks (new: the reference keeps each table
 * in one tf.Variable, macr_mf/model.py:112-113; BASELINE configs[4] = 10 M x 1 M rows,
 * d = 128 on 8 GPUs, where the dense Adam pass alone streams 33.8 GB per step).
 * Same arithmetic as macr_mf_train_step (all loss kinds; NORMALBCE has no (B,B) term and no branch vectors:
 * macr_shard_forward only prepares indices, macr_shard_bxb is not called, *branch_bytes comes back 0); the three
 * exchanges between the calls are the host's (torch.distributed / RCCL over xGMI,
 * macr_amd/sharded_train.py):
 *   macr_shard_gather    rows3[3][B][d] = the batch rows THIS rank owns, zero elsewhere      -> all-reduce(sum) rows3
 *                        Ownership of a table: local row l is global row lo + l * stride, l < n_loc (stride 1 = a
 *                        contiguous range; stride = W, lo = rank = interleaved: hot low ids spread over the ranks)
 *   macr_shard_forward   dots and branch factors of the whole batch (every rank, from rows3)
 *   macr_shard_bxb       this rank's row blocks of the (B,B) term into zeroed partials; *partials /
 *                        *partial_bytes = the fp32 region to sum over the ranks   -> all-reduce(sum)
 *   macr_shard_backward  losses[3] and the gradient rows of the whole batch (staging buffer inside
 *                        the workspace); *branch_grads / *branch_bytes = partial rows of dw, dw_user
 *                        (replicated work; broadcast rank 0's so w, w_user stay bit-identical)
 *   macr_shard_apply     references to this rank's rows are radix-sorted by row; the dense Adam pass over the
 *                        local shards sums each row's staged gradient rows itself (rows with more than 16
 *                        references through gP/gQ), and updates w, w_user
 * All five take the SAME workspace (>= macr_shard_workspace_bytes(B, d), 256-B aligned,
 * contents preserved between the calls of one step); adam_pow as in macr_mf_train_step.
 * -------------------------------------------------------------------------*/
size_t macr_shard_workspace_bytes(int B, int d);
int macr_shard_gather(int B, int d, const float *P_loc, int u_lo, int u_stride, int n_users_loc, const float *Q_loc,
                      int i_lo, int i_stride, int n_items_loc, const int32_t *u, const int32_t *i, const int32_t *j,
                      float *rows3, void *stream);
int macr_shard_forward(int loss_kind, int B, int d, const float *rows3, const float *w, const float *wu,
                       void *workspace, size_t workspace_bytes, void *stream);
int macr_shard_bxb(int B, int d, int rank, int world, void **partials, size_t *partial_bytes,
                   void *workspace, size_t workspace_bytes, void *stream);
int macr_shard_backward(int loss_kind, int B, int d, const float *rows3, const float *w, const float *wu,
                        float *adam_pow, const macr_hyper *hp, float *losses, void **branch_grads,
                        size_t *branch_bytes, void *workspace, size_t workspace_bytes, void *stream);
/* The SPLIT step (branch losses): rank r runs forward and backward only for the positions [t0, t1) of the batch whose (B,B) row
 * blocks it evaluates (macr_shard_slice), on rows their owners sent it, and returns the gradient rows to the owners -- two
 * all-to-alls of 3B/W rows per rank instead of an all-reduce of 3B rows (macr_amd/sharded_train.py::RowShardedMF.step_split).
 *   macr_shard_forward_slice   rows3_slice (dev) fp32[3][n][d]: user / positive / negative rows of positions t0 .. t0+n-1; returns
 *                              a region of the workspace (zero outside the slice) whose SUM over the ranks is the forward state
 *                              of the whole batch -- sum it (one all-reduce) before macr_shard_bxb
 *   macr_shard_backward_slice  after the partials of macr_shard_bxb were summed: gradient rows of the slice into stage_slice
 *                              (dev) fp32[3][n][d], losses of the whole batch, this slice's share of the branch-vector gradient
 *                              rows (returned region: sum over the ranks)
 *   macr_shard_stage           where macr_shard_apply reads gradient rows: (dev) fp32[3][B][d] inside the workspace, row
 *                              role * B + t; the caller fills the rows of the references this rank owns */
int macr_shard_slice(int B, int d, int rank, int world, int *t0, int *t1);
int macr_shard_forward_slice(int loss_kind, int B, int d, int t0, int n, const float *rows3_slice, const float *w,
                             const float *wu, void **region, size_t *region_bytes, void *workspace,
                             size_t workspace_bytes, void *stream);
int macr_shard_backward_slice(int loss_kind, int B, int d, int t0, int n, const float *rows3_slice, const float *w,
                              const float *wu, float *adam_pow, const macr_hyper *hp, float *losses,
                              float *stage_slice, void **branch_grads, size_t *branch_bytes, void *workspace,
                              size_t workspace_bytes, void *stream);
int macr_shard_stage(int B, int d, float **stage, void *workspace, size_t workspace_bytes);
/* macr_shard_route (abi 13): the routing tables of the split step from the batch alone, ONE launch (one workgroup; world <= 16).
 * References are numbered role * B + t (role 0 / 1 / 2 = user / positive / negative row of batch position t); owner(ref) = the rank
 * that holds the row, dest(ref) = the rank whose slice of positions (macr_shard_slice) holds t.  The same tables on every rank:
 *   bounds_u, bounds_i (dev) int32[world] first row NOT owned by rank q (contiguous ranges); NULL = interleaved rows, owner = row % world
 *   slice_end          (dev) int32[world] end t1 of rank q's slice
 *   counts             (dev) int32[world * world]  counts[q * world + p] = references owned by q whose position lies in p's slice:
 *                      row q / column p are this rank's send / receive split sizes of the two all-to-alls
 *   send_ref           (dev) int32[3B]  first n_send entries: the references THIS rank owns, ordered by (dest, reference)
 *   recv_ref           (dev) int32[3B]  first n_recv entries: the references of THIS rank's slice, ordered by (owner, reference)
 * No counterpart in the reference (one process, macr_mf/train.py:340); replaces RowShardedMF.route's bucketize / bincount / argsort. */
int macr_shard_route(int B, int world, int rank, const int32_t *u, const int32_t *i, const int32_t *j,
                     const int32_t *bounds_u, const int32_t *bounds_i, const int32_t *slice_end, int32_t *counts,
                     int32_t *send_ref, int32_t *recv_ref, void *stream);
int macr_shard_apply(int loss_kind, int B, int d, int n_users_loc, int n_items_loc, int u_lo, int u_stride, int i_lo,
                     int i_stride, const int32_t *u, const int32_t *i, const int32_t *j,
                     float *P_loc, float *Q_loc, float *w, float *wu,
                     float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu, float *vwu,
                     float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ, const macr_hyper *hp,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Lazy dense Adam: the dense pass of tf.train.AdamOptimizer (macr_mf/model.py:74,:95 -- every row of P and Q moves every
 * step, touched by the batch or not) blocked in TIME (new; the reference has one device and small tables).
 * A row without gradient moves by a recurrence of its own state and the step's lr_t (m <- b1 m, v <- b2 v,
 * theta <- theta - lr_t m / (sqrt(v) + eps)), so K such steps can be applied in registers in one trip to memory:
 * every row carries a STAMP (Adam steps received), the library keeps the lr_t of the last 256 steps, and a step
 *   - updates the rows its batch touched (their missing steps without gradient, then this step with it),
 *   - sweeps one K-th of every table up to the step (K = period),
 *   - leaves every other row alone: 24*d bytes per row every K steps instead of every step.
 * Rows are read through macr_shard_gather_lazy / macr_lazy_rows (brought to the current step in registers, nothing
 * written); macr_lazy_flush brings every row to the current step -- the tables then hold, bit for bit, what the
 * per-step dense pass leaves (call it before anything else reads P, Q or the slots).  The arithmetic per row and step
 * is the dense pass's; at BASELINE configs[4] (11 M rows, d = 128) the pass goes from 33.8 GB of HBM traffic per step
 * to the VALU time of the same arithmetic.
 *   state   (dev) MACR_LAZY_STATE_BYTES, zero-filled once: step counter + ring of lr_t
 *   stampP/Q (dev) uint32[rows of the (local) table], zero-filled once
 *   period  1 .. MACR_LAZY_MAX_PERIOD (1 = every row every step, through the lazy kernels)
 * -------------------------------------------------------------------------*/
#define MACR_LAZY_STATE_BYTES 1040
#define MACR_LAZY_MAX_PERIOD  64
typedef struct macr_lazy_adam {
    void     *state;
    uint32_t *stampP;
    uint32_t *stampQ;
    int       period;
} macr_lazy_adam;

/* macr_mf_train_step / macr_mf_train_flush with the lazy pass in place of the dense one (the (B,B) losses; same arguments +
 * lazy; stampP / stampQ: uint32[n_users] / uint32[n_items]).  Use ONE form for all calls of a deferred sequence (from the
 * first MACR_STEP_DEFER call to the flush or the call that completes it): the step counter advances in every call of the
 * lazy form.  In deferred mode the pass riding in a step's (B,B) launch updates the rows of the previous and of the current
 * batch (flag bit 0: gradient pending, bit 1: set by this step's forward kernel) and the step's K-th of the tables; the
 * forward kernel brings the rows it gathers up to date in registers.  A call that completes its step (no MACR_STEP_DEFER)
 * and macr_mf_train_flush_lazy end with every row at the current step: P, Q and the slots are then what the dense form
 * leaves, bit for bit -- between the calls of a deferred sequence they are NOT. */
int macr_mf_train_step_lazy(int loss_kind, int B, int d, int n_users, int n_items,
                            const int32_t *u, const int32_t *i, const int32_t *j,
                            float *P, float *Q, float *w, float *wu,
                            float *mP, float *vP, float *mQ, float *vQ,
                            float *mw, float *vw, float *mwu, float *vwu,
                            float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                            float *adam_pow, const macr_hyper *hp,
                            float *losses, int flags, const macr_lazy_adam *lazy,
                            void *workspace, size_t workspace_bytes, void *stream);
int macr_mf_train_flush_lazy(int loss_kind, int B, int d, int n_users, int n_items,
                             float *P, float *Q, float *w, float *wu,
                             float *mP, float *vP, float *mQ, float *vQ,
                             float *mw, float *vw, float *mwu, float *vwu,
                             float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                             const macr_hyper *hp, const macr_lazy_adam *lazy,
                             void *workspace, size_t workspace_bytes, void *stream);

/* macr_shard_gather on lazily updated shards (mP .. vQ: the slots of the local tables) */
int macr_shard_gather_lazy(int B, int d, const float *P_loc, const float *mP, const float *vP, int u_lo, int u_stride,
                           int n_users_loc, const float *Q_loc, const float *mQ, const float *vQ, int i_lo, int i_stride,
                           int n_items_loc, const int32_t *u, const int32_t *i, const int32_t *j, const macr_hyper *hp,
                           const macr_lazy_adam *lazy, float *rows3, void *stream);
/* out (dev) fp32[n][d]: row rows[k] (dev int32, local index; < 0: a zero row) of one lazily updated table as of the current step */
int macr_lazy_rows(long long n, int d, const int32_t *rows, const float *theta, const float *m, const float *v,
                   const uint32_t *stamp, const void *state, const macr_hyper *hp, float *out, void *stream);
/* macr_shard_apply with the lazy pass in place of the dense one (same arguments + lazy) */
int macr_shard_apply_lazy(int loss_kind, int B, int d, int n_users_loc, int n_items_loc, int u_lo, int u_stride, int i_lo,
                          int i_stride, const int32_t *u, const int32_t *i, const int32_t *j,
                          float *P_loc, float *Q_loc, float *w, float *wu,
                          float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu, float *vwu,
                          float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ, const macr_hyper *hp,
                          const macr_lazy_adam *lazy, void *workspace, size_t workspace_bytes, void *stream);
/* every row of both tables brought to the current step (n_rows_p / n_rows_q: rows of the local tables) */
int macr_lazy_flush(int d, long long n_rows_p, long long n_rows_q, float *P, float *Q, float *mP, float *vP, float *mQ,
                    float *vQ, const macr_hyper *hp, const macr_lazy_adam *lazy, void *stream);

/* ---------------------------------------------------------------------------
 * Device-side sampler of (user, positive, negative) triples (new).
 * Same distribution as Data.sample of the reference (macr_mf/load_data.py:543-566,
 * macr_lightgcn/utility/load_data.py:174-212): B distinct users from the pool (with
 * replacement when B > n_pool), a uniform positive from the user's train list (item 0
 * when empty), a uniform negative outside it.  NOT the reference's random stream: a
 * batch is a pure function of (seed, step).
 *   pool (dev, may be NULL = users 0..n_pool-1) int32[n_pool]: candidate user ids
 *   train_ptr (dev) int32[max_user+2], train_idx (dev): train items per user id, ascending
 *   out (dev) int32[3*B] = users | pos_items | neg_items
 * -------------------------------------------------------------------------*/
int macr_sample_triples(uint64_t seed, uint64_t step, int B, int n_items, const int32_t *pool, int n_pool,
                        const int32_t *train_ptr, const int32_t *train_idx, int32_t *out, void *stream);
/* The batches of steps step0 .. step0 + n_steps - 1 in ONE launch: out [n_steps][3][B], batch k identical to what
 * macr_sample_triples(seed, step0 + k, ...) draws (the generator is keyed by (seed, step, triple)).  A training loop
 * that draws a few dozen batches ahead pays one launch per few dozen steps instead of 8 us per step.
 *   excl_ptr / excl_idx (dev, may be NULL = the lists the positives come from): per user id, ascending item ids a
 *   negative must avoid -- Data.sample_test of the LightGCN loader draws positives from the TEST lists and negatives
 *   outside test and train lists (utility/load_data.py:214-254). */
int macr_sample_triples_many(uint64_t seed, uint64_t step0, int n_steps, int B, int n_items, const int32_t *pool,
                             int n_pool, const int32_t *train_ptr, const int32_t *train_idx,
                             const int32_t *excl_ptr, const int32_t *excl_idx, int32_t *out, void *stream);

/* ---------------------------------------------------------------------------
 * SpMM plan (host side, built once per graph -- the adjacency never changes).
 * Interaction graphs have hub rows (items with 10^4..10^5 neighbours); the plan
 * cuts rows longer than 512 non-zeros into work items of 512 so that no single
 * wavefront serialises a hub; the piece of a hub row that finishes last sums the
 * partial rows in fixed order (deterministic whichever piece that is).
 *   rowptr_host (HOST) int32[N+1]
 *   col_host / val_host (HOST, both or neither) int32 / fp32 [nnz]: with them the plan also carries the ENTRY STREAM of
 *               the dense layers -- the matrix once more in row order, every row followed by an end marker that makes
 *               the wave gather the row's running sum, cut into chunks of a few hundred entries (k_spmm_stream:
 *               no per-row descriptor, no per-row launch; rows above 512 entries in pieces of 255).  The device copy
 *               must be 64-byte aligned.  Without them every layer runs the one-wavefront-per-row kernel.
 *   plan_host   (HOST) >= macr_spmm_plan_bytes(...) bytes, written by
 *               macr_spmm_plan_build; the caller uploads a copy to the device and
 *               passes BOTH pointers (device copy for the kernels, host copy for the
 *               launch geometry) to the LightGCN entry points.  Passing NULL for both
 *               selects the plain one-wavefront-per-row kernel.
 * -------------------------------------------------------------------------*/
size_t macr_spmm_plan_bytes(int N, const int32_t *rowptr_host, const int32_t *col_host, const float *val_host);
int    macr_spmm_plan_build(int N, const int32_t *rowptr_host, const int32_t *col_host, const float *val_host,
                            void *plan_host, size_t plan_bytes);
size_t macr_lgcn_work_floats(int N, int d, const void *plan_host);   /* size of `work` below, in floats */

/* ---------------------------------------------------------------------------
 * LightGCN propagation  E = mean(E0, A E0, ..., A^L E0)   (L = n_layers)
 * Replaces _create_lightgcn_embed (macr_lightgcn/LightGCN.py:288-309): the
 * 100-fold tf.sparse_tensor_dense_matmul loop (:297-305) + stack/reduce_mean
 * (:306-307).  A is the `pre` adjacency D^-1/2 A D^-1/2 in CSR
 * (utility/load_data.py:112-121), N = n_users + n_items rows.
 *   rowptr (dev) int32[N+1], col (dev) int32[nnz], val (dev) fp32[nnz]
 *   plan_dev (dev) / plan_host (HOST): the SpMM plan, or both NULL
 *   E0 (dev) fp32[N*d] in, E (dev) fp32[N*d] out
 *   work (dev) fp32[macr_lgcn_work_floats(N, d, plan_host)] scratch.  ZERO-FILL IT ONCE before its first use: with a
 *            plan it holds the hub rows' arrival counters, which every call leaves at zero again.
 * The same call is the backward pass: feed dE and the TRANSPOSED adjacency (A itself when symmetric), get dE0.
 * -------------------------------------------------------------------------*/
int macr_lgcn_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col,
                        const float *val, const void *plan_dev, const void *plan_host,
                        const float *E0, float *E, float *work, void *stream);

/* One LightGCN training step.  Replaces sess.run([opt_X, loss_X, mf_loss_X,
 * emb_loss_X, reg_loss_X]) of macr_lightgcn/LightGCN.py:598-607: propagation,
 * gathers on the propagated table (:145-150), loss (:415-429 or :495-532) with
 * the regulariser on the ego rows (:525-527), dense gradients through the
 * propagation, Adam (:186 / :201).
 *   T (dev) fp32[N*d] = [user_embedding ; item_embedding], updated in place
 *   mT,vT Adam slots;  workspace (dev) >= macr_lgcn_train_workspace_bytes(B,N,d,plan_host) bytes, ZERO-FILLED ONCE
 *          before its first use (the batch-row flags and the hub rows' arrival counters are zero
 *          between steps: every step leaves them that way)
 *   losses (dev) fp32[3] = {loss, mf_loss, emb_loss}
 *   flags  0, or MACR_STEP_LOSS_ONLY: sess.run([loss_X, mf_loss_X, emb_loss_X]) without opt_X -- the
 *          reference's per-log-interval "test loss" pass (LightGCN.py:799-819, train_thread_test :620-647):
 *          propagation + forward of the loss; T, w, wu, the slots and adam_pow are left untouched.
 * -------------------------------------------------------------------------*/
size_t macr_lgcn_train_workspace_bytes(int B, int N, int d, const void *plan_host);

int macr_lgcn_train_step(int loss_kind, int B, int d, int n_users, int n_items, int n_layers,
                         const int32_t *rowptr, const int32_t *col, const float *val,
                         const void *plan_dev, const void *plan_host,
                         const int32_t *u, const int32_t *i, const int32_t *j,
                         float *T, float *w, float *wu, float *mT, float *vT,
                         float *mw, float *vw, float *mwu, float *vwu,
                         float *adam_pow, const macr_hyper *hp,
                         float *losses, int flags, void *workspace, size_t workspace_bytes, void *stream);

/* The same step for an adjacency that is NOT symmetric (abi 13): --adj_type norm / gcmc / mean of macr_lightgcn/LightGCN.py:667-678
 * are row-normalised, D^-1 A (utility/load_data.py:95-164), and the gradient of A E is A^T dE (tf.gradients of
 * tf.sparse_tensor_dense_matmul, LightGCN.py:301).  rowptr_t / col_t / val_t / plan_t_*: the TRANSPOSED adjacency in CSR with a
 * plan of its own (both plans or neither); the forward propagation uses A, the backward propagation A^T.  macr_lgcn_train_step is
 * this call with A in both places.  workspace >= macr_lgcn_train_workspace_bytes_t(B, N, d, plan_host, plan_t_host). */
size_t macr_lgcn_train_workspace_bytes_t(int B, int N, int d, const void *plan_host, const void *plan_t_host);
int macr_lgcn_train_step_t(int loss_kind, int B, int d, int n_users, int n_items, int n_layers,
                           const int32_t *rowptr, const int32_t *col, const float *val,
                           const void *plan_dev, const void *plan_host,
                           const int32_t *rowptr_t, const int32_t *col_t, const float *val_t,
                           const void *plan_t_dev, const void *plan_t_host,
                           const int32_t *u, const int32_t *i, const int32_t *j,
                           float *T, float *w, float *wu, float *mT, float *vT,
                           float *mw, float *vw, float *mwu, float *vwu,
                           float *adam_pow, const macr_hyper *hp,
                           float *losses, int flags, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * out[r] = sigmoid(rows[r] . w)   -- the test-time branch factors
 * tf.nn.sigmoid(tf.matmul(e, w)) of macr_mf/model.py:194-196,:199.
 * idx (dev, may be NULL) selects rows: out[r] = sigmoid(rows[idx[r]] . w).
 * -------------------------------------------------------------------------*/
int macr_branch_sigmoid(const float *rows, const int32_t *idx, int n, int d, const float *w,
                        float *out, void *stream);
/* Two of them in one launch: the item and the user factors of an evaluation (same arithmetic per row). */
int macr_branch_sigmoid2(int d, const float *rows_a, const int32_t *idx_a, int n_a, const float *w_a, float *out_a,
                         const float *rows_b, const int32_t *idx_b, int n_b, const float *w_b, float *out_b, void *stream);

/* ---------------------------------------------------------------------------
 * Fused full-catalogue scoring + train-item masking + top-K: never
 * materialises the (U,N) score matrix.
 * Replaces sess.run(model.batch_ratings | model.rubi_ratings_both,
 * {users: batch, pos_items: range(N)}) (macr_mf/train.py:224-251,
 * utility/batch_test.py:50-93) + the candidate filtering and ranking of
 * macr_mf/train.py:119-138 / batch_test.py:124-134 + tools.h:13-33.
 *
 *   users_tab (dev) fp32[*, d], user_ids (dev) int32[U]: row user_ids[q] of
 *              users_tab is query q (user_ids may be NULL = rows 0..U-1)
 *   items     (dev) fp32[n_local*d]: the LOCAL item shard; global item id of
 *              local row t is item_offset + t
 *   sig_u     (dev) fp32[U]  sigmoid(e_u . w_user)  per query  (RUBI_BOTH only)
 *   sig_i     (dev) fp32[n_local] sigmoid(e_i . w)  per local item (RUBI_BOTH)
 *   c         the constant `rubi_c` set by update_c (model.py:313)
 *   c_dev     (dev, may be NULL) fp32[1]: if given, the kernels read c from it at RUN time and ignore `c`: a
 *              call sequence captured into a hipGraph then serves every c of a sweep (macr_mf/tune.py:545-578,
 *              LightGCN_tune.py:852-870) -- the caller rewrites the scalar between replays
 *   mask_ptr  (dev) int32[U+1], mask_idx (dev) int32[*]: per query, ascending
 *              GLOBAL item ids to exclude (the user's train items); may be NULL
 *   n_splits  >= 1: number of (U,K) result lists in out_val/out_idx; 0 = let the
 *              library choose (macr_score_topk_splits(), 1 since ABI 5).  The ranking
 *              balances its own grid over (query block, item tile) ranges; the merged
 *              result of the whole shard is list 0, further lists hold only padding.
 *              (n_splits > 1 still shapes the grid of the fallback kernel, below.)
 *   seed_idx  (dev, may be NULL) int32[U*MACR_SEED_WIDTH]: per query, GLOBAL ids of items that ranked high for it before --
 *              what seed_out held after the previous call for the same queries (an evaluator ranks the same users against
 *              slowly moving tables every epoch).  When given, the per-query threshold is the K-th largest EXACT current
 *              score of its seeds (a valid lower bound of the K-th best score for any set of distinct unmasked items; the
 *              result does not depend on the seeds, only the time does) and the sampling pass and its selection kernel are
 *              skipped.  Seeds are checked on the device: an id that is -1, outside this shard, masked or repeated does
 *              not count, and a query left with fewer than K seeds lists every unmasked item.  Seeds from a ranking the
 *              tables have moved far away from cost a repair round (below), never a wrong result.
 *   seed_out  (dev, may be NULL, may be seed_idx itself) int32[U*MACR_SEED_WIDTH] <- the best MACR_SEED_WIDTH candidates the
 *              selection saw per query, best first, -1 padded (the first K of them are out_idx): the next call's seed_idx.
 *   stats     (dev, may be NULL) int32[2], written by the call: [0] blocks of 256 queries the repair round listed again,
 *              [1] 1 if the exact fallback kernel ran.  A caller that seeds uses [0] to tell stale seeds from good ones.
 *   out_val (dev) fp32 [n_splits*U*K], out_idx (dev) int32 [n_splits*U*K]:
 *              per list, per query: K (score,id) pairs, score descending, ties
 *              by ascending id, unused slots (-inf, -1).  Feed to macr_topk_merge
 *              (with the other shards' lists on several GPUs).
 *   workspace (dev) >= macr_score_topk_workspace_bytes(U, n_local, d) bytes, 256-B
 *              aligned: per-query thresholds, the (item tile, query) mask bitmap, the
 *              per-(slot,query) candidate lists of the fixed-threshold stream (512 or
 *              1024 keys of 8 bytes each) and the two-term bf16 copies of the operands (as
 *              many bytes as the fp32 rows: the bf16 candidate filter, below).  Contents
 *              need not be initialised or preserved.
 * Score: NORMAL e_u.e_i ; RUBI_BOTH ((e_u.e_i - c) * sig_i) * sig_u ; RUBI,
 * DIRECT_MINUS, DIRECT_MINUS_BOTH as listed at the MACR_SCORE_* constants (every
 * operation rounds on its own, in the order of the reference's expression), the dot
 * product being a k-ascending fp32 fma chain (gfx950 fp32 MFMA arithmetic).
 * sig_i is needed by every kind but NORMAL, sig_u by RUBI_BOTH and DIRECT_MINUS_BOTH.
 * d in {32,64,128,256}; 1 <= K <= MACR_MAX_TOPK.
 * Launches (without seeds): a sampling pass over every 8th item (per-query lower bound tau of
 * the K-th best score), the listing pass over all tiles (items scoring >= tau go to
 * per-query candidate lists), a selection kernel (exact top K of each list); then the
 * REPAIR round for queries whose list overflowed (threshold too loose): their K-th listed
 * score becomes their threshold and the listing pass + selection run again for the blocks
 * of 256 queries that hold them, which then share the whole list buffer (launches that
 * return at once when nothing overflowed); and a fallback launch of the running-top-K
 * kernel whose blocks return at once unless a list overflowed again (exact for any input).
 * With seeds: one kernel
 * (k_tau_seed) instead of the sampling pass and its selection, and a listing pass that gives a
 * block of queries up to the repair round as soon as its lists grow faster than a usable
 * threshold allows (stale seeds).  No host synchronisation.
 * -------------------------------------------------------------------------*/
int    macr_score_topk_splits(int U, int n_local, int d);
int    macr_score_topk_uses_seeds(int U, int n_local, int d);   /* 0: this shape lists every unmasked item, seed_idx is ignored */
size_t macr_score_topk_workspace_bytes(int U, int n_local, int d);

/* How the listing pass of macr_score_topk forms its candidate lists.  The ranking returned is the fp32 ranking either
 * way, bit for bit (scores, ids, tie order): a filter only decides which (query, item) pairs get a closer look.
 *   MACR_EVAL_FILTER_F32   the (U, N) product on the fp32 matrix cores; a listed score is the final score.
 *   MACR_EVAL_FILTER_BF16  the product on two-term bf16 splits of the operands (hi*hi + hi*lo + lo*hi: 5x fewer
 *                          matrix-core cycles), every score compared with the query's threshold LESS a rigorous bound of
 *                          the error ((3.2 * 2^-16 + 8 d * 2^-24) |u| max|q| + roundings: 7.9e-5 at d = 64, 1.71e-4 at
 *                          d = 256), in the sampling pass, the listing pass and the
 *                          repair round; the selection takes each query's 64 best candidates by bf16 score, checks that
 *                          nothing else can belong to the top K, re-computes the fp32 score of those that can and ranks
 *                          them; a query that fails the check has all its listed candidates re-scored in fp32.
 *   MACR_EVAL_FILTER_F16   (abi 14) the product on ONE fp16 number per operand (v_mfma_f32_32x32x16_f16: a third of the
 *                          bf16 filter's matrix-core work, half its operand bytes), margin (1.001 * 2^-10 + 8 d * 2^-24)
 *                          |u| max|q| + roundings (1.0e-3 at d = 64) -- 12x the bf16 filter's, still several times
 *                          below the gap between a query's K-th and 64th best score; the selection, the fp32 re-scoring
 *                          and the per-query fall-back are the bf16 filter's.  An operand outside fp16's range
 *                          (|x| > 65504) is clamped and voids the bound for its row: that query (every query, for an
 *                          item row) ends in the exact kernel.  macr_score_topk_sweep has no fp16 kernels and runs
 *                          its bf16 ones under this value.
 *   MACR_EVAL_FILTER_ENV   follow MACR_EVAL_FILTER=f32|bf16|f16 in the environment, f32 when unset.
 * An ARGUMENT of every ranking call (`filter`, second parameter) since abi 10: the library keeps no filter state (round 4's
 * process-wide macr_set_eval_filter is gone).  A first-round call and the repair-round call that finishes it take the
 * same value. */
#define MACR_EVAL_FILTER_ENV  0
#define MACR_EVAL_FILTER_F32  1
#define MACR_EVAL_FILTER_BF16 2
#define MACR_EVAL_FILTER_F16  3
/* OR into `filter` of macr_score_topk / macr_score_topk_first_round (abi 12): the head of `workspace` was initialised by
 * macr_score_topk_prologue for exactly this call, which then launches no initialisation of its own. */
#define MACR_EVAL_WS_READY    0x100
/* OR into `filter` together with MACR_EVAL_WS_READY and MACR_EVAL_FILTER_F16 (abi 15): macr_score_topk_prologue_prep has also
 * written the fp16 filter's operand copies for exactly this call (same score_kind, c, tables), which then converts nothing. */
#define MACR_EVAL_PREP_READY  0x200

int macr_score_topk(int score_kind, int filter, int U, int n_local, int d,
                    const float *users_tab, const int32_t *user_ids, const float *items,
                    const float *sig_u, const float *sig_i, float c, const float *c_dev,
                    const int32_t *mask_ptr, const int32_t *mask_idx, const uint32_t *mask_bits,
                    int item_offset, int K, int n_splits, const int32_t *seed_idx, int32_t *seed_out,
                    float *out_val, int32_t *out_idx, int32_t *stats,
                    void *workspace, size_t workspace_bytes, void *stream);

/* The FIRST ROUND of macr_score_topk alone, for a caller that reads results back after every ranking anyway (each test()
 * of macr_mf/train.py:162-311 / utility/batch_test.py:26-162 returns metrics to the host): same arguments, but neither the
 * repair round nor the exact fallback kernel is launched -- in the common case those five launches find nothing to do.
 * stats (required; device memory or device-visible host memory) receives {query blocks whose candidate lists overflowed
 * or whose seeds were stale, 0}.  stats[0] == 0: out_val / out_idx / seed_out are exactly what macr_score_topk returns.
 * Otherwise they are not the ranking yet: macr_score_topk_repair_round finishes it (or macr_score_topk starts over). */
int macr_score_topk_first_round(int score_kind, int filter, int U, int n_local, int d,
                    const float *users_tab, const int32_t *user_ids, const float *items,
                    const float *sig_u, const float *sig_i, float c, const float *c_dev,
                    const int32_t *mask_ptr, const int32_t *mask_idx, const uint32_t *mask_bits,
                    int item_offset, int K, int n_splits, const int32_t *seed_idx, int32_t *seed_out,
                    float *out_val, int32_t *out_idx, int32_t *stats,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ... and the REST of macr_score_topk for the rare call whose first round did not stand: the same arguments (seed_idx as
 * in the first-round call: it decides whether the re-listed query blocks are sampled first), the workspace exactly as
 * macr_score_topk_first_round left it and the same out_val / out_idx / seed_out buffers -- the repair round re-lists the
 * stats[0] query blocks and overwrites their rows; the exact fallback kernel follows if a list overflows again.
 * first_round + repair_round launch what macr_score_topk launches.  stats as macr_score_topk writes it. */
int macr_score_topk_repair_round(int score_kind, int filter, int U, int n_local, int d,
                    const float *users_tab, const int32_t *user_ids, const float *items,
                    const float *sig_u, const float *sig_i, float c, const float *c_dev,
                    const int32_t *mask_ptr, const int32_t *mask_idx, const uint32_t *mask_bits,
                    int item_offset, int K, int n_splits, const int32_t *seed_idx, int32_t *seed_out,
                    float *out_val, int32_t *out_idx, int32_t *stats,
                    void *workspace, size_t workspace_bytes, void *stream);

/* What an evaluation of the two-branch scores launches before its ranking, as ONE launch (abi 12): the branch factors
 * sig_i[n_local] = sigmoid(items . w_item) and, when sig_u is given, sig_u[U] = sigmoid(users_tab[user_ids] . w_user)
 * (macr_mf/model.py:141-142,:199-201; macr_branch_sigmoid's arithmetic, bit for bit), and the initialisation of the head of
 * `workspace` that macr_score_topk(_first_round) would start with.  filter, U, n_local, d, K as in the ranking call that
 * follows on the same stream with MACR_EVAL_WS_READY in its filter; seeded_first_round != 0: that call is
 * macr_score_topk_first_round with seed_idx != NULL.  sig_u, w_user may both be NULL (one-branch scores). */
int macr_score_topk_prologue(int filter, int U, int n_local, int d, int K, int seeded_first_round,
                             const float *items, const float *w_item, float *sig_i,
                             const float *users_tab, const int32_t *user_ids, const float *w_user, float *sig_u,
                             void *workspace, size_t workspace_bytes, void *stream);

/* The same and, in the same launch, the fp16 filter's operand copies for the ranking call that follows (abi 15): every row of
 * `items` and every query row is read ONCE -- for its branch factor (computed from the row in hand, macr_branch_sigmoid's
 * arithmetic bit for bit) and for its fp16 copy, norm and bias -- where macr_score_topk_prologue + the ranking call's own
 * conversion read them twice, cold after a log interval of training (the tables are what training has just rewritten).
 * That call takes filter = MACR_EVAL_FILTER_F16 | MACR_EVAL_WS_READY | MACR_EVAL_PREP_READY and the same score_kind, c / c_dev,
 * U, n_local, d, K, tables, workspace and stream.  K <= MACR_MAX_TOPK_FUSED; a catalogue small enough to list every item
 * (no filter runs there) is refused with MACR_E_UNSUPPORTED: use macr_score_topk_prologue. */
int macr_score_topk_prologue_prep(int score_kind, int U, int n_local, int d, int K, int seeded_first_round,
                                  const float *items, const float *w_item, float *sig_i,
                                  const float *users_tab, const int32_t *user_ids, const float *w_user, float *sig_u,
                                  float c, const float *c_dev, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * The same ranking for SEVERAL values of c at once -- the c sweep of the tuners
 * (macr_mf/tune.py:545-578, macr_lightgcn/LightGCN_tune.py:852-870: test() once per
 * np.linspace(start, end, step) value).  c only enters the score epilogue, so ONE
 * listing pass (the MFMA work, the staging of the item tiles) serves up to
 * MACR_MAX_SWEEP values; per value there remain the sampling pass, the threshold, the
 * appends into its own candidate lists and the selection.
 *   c_dev     (dev) fp32[n_c], 1 <= n_c <= MACR_MAX_SWEEP, read at run time
 *   mask_bits (dev) the bitmap of macr_mask_bits_build (required when a mask is given)
 *   out_val/out_idx (dev) [n_c][U][K]: per value of c, per query, K (score,id) pairs as
 *              macr_score_topk writes them (one list per value: n_splits = 1)
 *   workspace >= macr_score_topk_sweep_workspace_bytes(U, n_local, d, n_c), 256-B aligned
 * score_kind: any kind that uses c (not MACR_SCORE_NORMAL).  Results per value are
 * identical to macr_score_topk with that c.  Follows the candidate filter (`filter`, as above):
 * with the bf16 filter the shared listing pass, the sampling passes and the selections are the
 * bf16 kernels.
 * -------------------------------------------------------------------------*/
#define MACR_MAX_SWEEP 4
size_t macr_score_topk_sweep_workspace_bytes(int U, int n_local, int d, int n_c);
int macr_score_topk_sweep(int score_kind, int filter, int U, int n_local, int d,
                          const float *users_tab, const int32_t *user_ids, const float *items,
                          const float *sig_u, const float *sig_i, int n_c, const float *c_dev,
                          const int32_t *mask_ptr, const int32_t *mask_idx, const uint32_t *mask_bits,
                          int item_offset, int K, float *out_val, int32_t *out_idx,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The train-item mask as the ranking kernels read it: mask_bits[tile][query] has bit (i % 32)
 * set when the query masks item 32*tile + i of the shard.  The mask of an evaluator never
 * changes during training (macr_mf/train.py:119-138 filters the same train lists every
 * epoch), so it is built once and passed to every macr_score_topk call of the same
 * (queries, shard); with mask_bits == NULL macr_score_topk builds it per call into its
 * workspace. */
size_t macr_mask_bits_bytes(int U, int n_local);
int    macr_mask_bits_build(int U, int n_local, const int32_t *mask_ptr, const int32_t *mask_idx,
                            int item_offset, uint32_t *mask_bits, void *stream);

/* Dense scores for callers that want the matrix itself (the literal
 * sess.run(model.rubi_ratings_both, ...) -> (U,N) fp32 contract). */
int macr_score_matrix(int score_kind, int U, int n_local, int d,
                      const float *users_tab, const int32_t *user_ids, const float *items,
                      const float *sig_u, const float *sig_i, float c, const float *c_dev,
                      float *out_scores, void *stream);

/* ---------------------------------------------------------------------------
 * Top-K column indices of every row of a score matrix.
 * Replaces  void c_top_k_array_index(float *scores_pt, int columns_num,
 *            int rows_num, int top_k, int thread_num, int *rankings_pt)
 * (macr_lightgcn/evaluator/cpp/include/tools.h:24; binding in
 * apt_evaluate_foldout.pyx:11-13).  scores (dev) fp32[rows*cols]; -inf entries
 * (masked train items, batch_test.py:129) rank last; ties by ascending index
 * (the reference leaves tie order to std::partial_sort_copy).
 * out_idx (dev) int32[rows*K], out_val (dev, may be NULL) fp32[rows*K].  Any K >= 1 (the reference's function has no
 * bound on top_k; positions past the number of columns hold -1 / -inf); workspace (dev, may be NULL up to
 * K = MACR_MAX_TOPK) >= macr_topk_scores_workspace_bytes(rows, K).
 * -------------------------------------------------------------------------*/
size_t macr_topk_scores_workspace_bytes(int rows, int K);      /* 0 for K <= MACR_MAX_TOPK */
int macr_topk_scores(const float *scores, int cols, int rows, int K,
                     int32_t *out_idx, float *out_val, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * Merge W lists of K (score,id) pairs per query into one (new; also the merge
 * step after the RCCL all-gather of per-shard top-K).  vals/idxs (dev)
 * [W*U*K]; out (dev) [U*K]; out_cnt (dev) int32[U] = number of real candidates.
 * If fill_mask_ptr != NULL, queries with fewer than K candidates are completed
 * with their masked ids in ascending order at score -inf (what ranking a
 * -inf-masked matrix yields, batch_test.py:124-134).
 * -------------------------------------------------------------------------*/
int macr_topk_merge(int W, int U, int K, const float *vals, const int32_t *idxs,
                    const int32_t *fill_mask_ptr, const int32_t *fill_mask_idx,
                    float *out_val, int32_t *out_idx, int32_t *out_cnt, void *stream);

/* ---------------------------------------------------------------------------
 * Fold-out metrics.  Replaces  void evaluate_foldout(int users_num,
 *   int *rankings, int rank_len, int **ground_truths, int *ground_truths_num,
 *   int thread_num, float *results)
 * (macr_lightgcn/evaluator/cpp/include/evaluate_foldout.h:115-118).
 * rankings (dev) int32[U*K]; ground truth as CSR (gt_ptr int32[U+1], gt_idx
 * ascending per query) instead of int**; results (dev) fp32[U*5*K] laid out
 * [precision | recall | ap | ndcg | mrr] x K prefixes per query.
 * hr_in_ap_slot != 0 additionally applies the caller-side rewrite of
 * macr_lightgcn/utility/batch_test.py:143-149: the ap block is replaced by
 * HR := 1[recall@k != 0].
 * -------------------------------------------------------------------------*/
int macr_metrics_foldout(int U, int K, const int32_t *rankings,
                         const int32_t *gt_ptr, const int32_t *gt_idx, float *results,
                         int hr_in_ap_slot, void *stream);

/* The same on the one sorted list per query macr_score_topk leaves (ids, -1 in unused slots), completed -- where a query has
 * fewer than K candidates -- with its masked ids in ascending order from the fill CSR (fill_ptr int32[U+1], fill_idx): what
 * macr_topk_merge's fill writes for the reference's -inf train items (batch_test.py:124-134), without that launch.
 * K <= MACR_MAX_TOPK. */
int macr_metrics_foldout_fill(int U, int K, const int32_t *rankings, const int32_t *fill_ptr, const int32_t *fill_idx,
                              const int32_t *gt_ptr, const int32_t *gt_idx, float *results, int hr_in_ap_slot, void *stream);

/* MF metrics (macr_mf/train.py:32-117, float64 like NumPy): out (dev) f64
 * [U*4*nK] = per query {precision, recall, ndcg, hit_ratio} x Ks.  cnt (dev,
 * may be NULL) int32[U] = length of each ranked list; NULL: a list's length is its
 * number of ids >= 0 (unused slots hold -1: what macr_topk_merge reports as its count),
 * so the one sorted list per query that macr_score_topk leaves needs no merge first. */
int macr_metrics_mf(int U, int Kmax, const int32_t *rankings, const int32_t *cnt,
                    const int32_t *gt_ptr, const int32_t *gt_idx,
                    const int32_t *Ks /*host*/, int nK, double *out, void *stream);

/* macr_metrics_mf and the means over the query users in ONE launch (abi 12): what macr_metrics_mf followed by macr_colmean
 * returns for the "/ n_test_users" accumulation of macr_mf/train.py:286-290, without the second launch and the (U,4,nK) trip
 * through memory.  mean f64[4*nK]: device memory or device-visible (pinned) host memory.  per_user (dev, may be NULL):
 * macr_metrics_mf's out, for callers that want both.  The sum is a fixed two-level tree over blocks of 64 queries
 * (a function of U and nK only: deterministic; it differs from macr_colmean's tree in the last bits).  workspace (dev, 8-byte
 * aligned, macr_metrics_mf_mean_workspace_bytes): its first 4 bytes are a ticket that must be ZERO on entry and is zero on
 * return (zero-fill the buffer once); one call at a time per workspace. */
size_t macr_metrics_mf_mean_workspace_bytes(int U, int nK);
int macr_metrics_mf_mean(int U, int Kmax, const int32_t *rankings, const int32_t *cnt,
                         const int32_t *gt_ptr, const int32_t *gt_idx,
                         const int32_t *Ks /*host*/, int nK, double *per_user, double *mean,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Column means of a (rows, cols) matrix in float64 (deterministic tree):
 * the "/ n_test_users" accumulation of macr_mf/train.py:286-290 and the
 * np.mean(all_result, axis=0) of batch_test.py:151.  in_is_f32 selects input
 * type; out f64[cols]: device memory or device-visible (pinned) host memory. */
int macr_colmean(const void *in, int in_is_f32, int rows, int cols, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MACR_HIP_H */
