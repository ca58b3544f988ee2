// Training-step kernels of the MACR hot path for gfx950 (MI355X).
//
// One step =  pair_fwd  ->  bxb (rubibceboth only)  ->  pair_bwd  ->  adam_dense
// (normalbce fuses fwd+bwd into pair_normal).  In deferred mode (MACR_STEP_DEFER /
// MACR_STEP_PENDING) the dense Adam pass of step t runs as extra blocks of the bxb
// launch of step t+1 and pair_fwd looks one update ahead in registers:
//   pair_fwd<PENDING>  ->  bxb+adam  ->  pair_bwd          (macr_mf_train_flush ends it)
// What each kernel replaces in the reference is cited at the kernel; the arithmetic
// follows SURVEY.md appendix A.
//
// Data layout: embedding tables are row-major fp32 [rows][d]; one row is
// 4*LPR floats and is always touched as LPR lanes x float4 (a 256-B row at
// d=64 is one coalesced 16-lane access).  A wave therefore works on 64/LPR
// rows at once and reduces dot products inside each LPR-lane group with
// cross-lane shuffles.
#include "common.hpp"
#include "refsort.hpp"

#include <type_traits>

namespace macr {

// Timing probes (-DMACR_ABL_*: wrong results, never defined in the product build; the measurements are under profiles/r02_ablations).
#ifdef MACR_ABL_NOATOMIC
#define MACR_ATOMIC_ADD(p, v) (*(p) = (v))
#else
#define MACR_ATOMIC_ADD(p, v) unsafeAtomicAdd((p), (v))
#endif

// ----------------------------------------------------------------------------
// Small per-step scalar block living at the start of the workspace.
// ----------------------------------------------------------------------------
// Partial-sum slots written by the pair kernels (one set per block), reduced by adam_dense block 0.
// slot 0: sum of squares (regulariser)   1: L_item terms   2: L_user terms   3: per-pair BCE (normalbce)
constexpr int kPartStride = 4;

template <int LPR>
struct RowGroup {
    static constexpr int kRowsPerWave = kWave / LPR;
    static constexpr int kRowsPerBlock = 256 / LPR;
    int sub;    // lane inside the group: owns floats [4*sub, 4*sub+4) of the row
    int slot;   // row slot inside the block
    __device__ RowGroup() {
        sub = threadIdx.x % LPR;
        slot = threadIdx.x / LPR;
    }
};

// ----------------------------------------------------------------------------
// adam_dense: tf.train.AdamOptimizer as TF 1.14 applies it to embedding tables
// (macr_mf/model.py:74,:95): EVERY row decays m,v and moves every
// step; rows touched by the batch additionally consume their summed gradient.
//   m=b1*m+(1-b1)g; v=b2*v+(1-b2)g^2; theta-=lr_t*m/(sqrt(v)+eps)
// Streaming pass, float4 per lane: 24*d bytes per row (read+write theta,m,v) plus a
// 4-byte row flag; the gradient row is read (and re-zeroed) only when flagged.
// The same block routine serves the stand-alone kernel and the Adam blocks that ride in
// the bxb launch (deferred mode).  Branch vectors (one row each) take their gradient from
// the kBranchSlots partial rows pair_bwd adds into, and leave them zero.
// ----------------------------------------------------------------------------
struct AdamSeg {
    float *theta, *m, *v, *g;
    int n_parts;             // > 0: g holds n_parts partial rows (stride part_stride floats) to be summed
    int part_stride;
    int32_t *touched;        // NULL: gradient is dense, always read, left untouched
    uint32_t *stamp;         // lazy pass only (k_adam_lazy): Adam steps the row has received
    const uint32_t *sv;      // INDEXED pass (large batches): sorted reference list (values = staging rows) and the staging
    const float *stage;      //   buffer; a flag with a length field names a row's references instead of a row of g
    long long n_vec;         // number of float4 in the segment
    long long first_block;   // first block index serving this segment
};
struct LazyState;
struct AdamArgs {
    AdamSeg seg[4];
    int n_seg;
    int lpr;                 // float4 per row (8, 16, 32 or 64)
    int lpr_shift;           // log2(lpr): row of a float4 = index >> lpr_shift (a 64-bit division per access otherwise)
    float b1, b2, eps;
    const LazyState *lazy = nullptr;   // lazy pass riding in the (B,B) launch (adam_lazy_scan_block): step counter + lr_t ring
    int period = 1;
};
struct LossArgs {
    const float *part; int n_part;       // pair-kernel partials  [n_part][4]
    const float *part2; int n_part2;     // second partial set (LightGCN ego regulariser), slot 0 only
    const float *lpart; int n_lpart;     // bxb loss partials
    int kind, B, batch_size_cfg;
    float alpha, beta, decay;
    float *losses;
};

#ifndef MACR_ADAM_ITERS
#define MACR_ADAM_ITERS 4
#endif
constexpr int kAdamIters = MACR_ADAM_ITERS;          // float4 per thread and array (2, 4, 8 measured: no difference)
constexpr int kAdamVecPerBlock = 256 * kAdamIters;
constexpr int kBranchSlots = 8;             // partial rows of the branch-vector gradients (pair_bwd adds, Adam consumes)

// The two moment updates are written as the fused forms the compiler chose for the dense pass (one product rounded, the
// other inside the fma) instead of `m*b1 + g*(1-b1)`: left to -ffp-contract the choice depends on the surrounding code, and
// the same row is updated in different kernels (the dense pass, the look-ahead of pair_fwd, the lazy passes) whose results
// must agree bit for bit.
__device__ __forceinline__ void adam4(float4 &th, float4 &m, float4 &v, const float4 gr, float lr_t, float b1,
                                      float b2, float eps) {
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    m.x = fmaf(m.x, b1, gr.x * omb1); m.y = fmaf(m.y, b1, gr.y * omb1); m.z = fmaf(m.z, b1, gr.z * omb1); m.w = fmaf(m.w, b1, gr.w * omb1);
    v.x = fmaf(v.x, b2, (gr.x * gr.x) * omb2); v.y = fmaf(v.y, b2, (gr.y * gr.y) * omb2);
    v.z = fmaf(v.z, b2, (gr.z * gr.z) * omb2); v.w = fmaf(v.w, b2, (gr.w * gr.w) * omb2);
    th.x = adam_update(th.x, m.x, v.x, lr_t, eps); th.y = adam_update(th.y, m.y, v.y, lr_t, eps);
    th.z = adam_update(th.z, m.z, v.z, lr_t, eps); th.w = adam_update(th.w, m.w, v.w, lr_t, eps);
}

// One block's share of the pass: block `blk` of the segment list.  s_red (256 float4) is only used by
// partial-row segments (PARTS).
// Row flags of the large-batch path (k_seg_reduce<LPR, true>): 0 = no gradient; 1 = the row of g holds it; otherwise
// (length << kRefShift) | position: the row's gradient is the sum of `length` (<= kRefRun) staging rows, the ones the sorted
// reference list names from `position` on.  An INDEXED pass sums them itself, in list order -- the order k_seg_reduce
// sums in, so the two paths give the same bits -- and the gradient never exists as a row in memory: nothing writes it,
// reads it back or clears it.
constexpr int kRefShift = 26;
constexpr uint32_t kRefMaxRefs = 1u << kRefShift;       // positions must fit below the length field

template <bool PARTS, bool INDEXED = false>
__device__ __forceinline__ void adam_block(const AdamArgs &a, long long blk, float lr_t, float4 *s_red) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.n_seg && blk >= a.seg[k].first_block) s = k;
    const AdamSeg sg = a.seg[s];
    const long long base = (blk - sg.first_block) * kAdamVecPerBlock + threadIdx.x;
    float4 gsum = make_float4(0, 0, 0, 0);
    if (PARTS && sg.n_parts > 0) {
        // branch-vector segment (one row): the whole block sums the per-block partial rows of pair_bwd
        const int sub = threadIdx.x & (a.lpr - 1), grp = threadIdx.x >> a.lpr_shift, ngrp = 256 >> a.lpr_shift;
        for (int k = grp; k < sg.n_parts; k += ngrp) {
            gsum = add4(gsum, ld4(sg.g + (size_t)k * sg.part_stride + 4 * sub));
            st4(sg.g + (size_t)k * sg.part_stride + 4 * sub, make_float4(0, 0, 0, 0));
        }
        s_red[threadIdx.x] = gsum;
        __syncthreads();
        gsum = make_float4(0, 0, 0, 0);
        if (threadIdx.x < a.lpr)
            for (int k = 0; k < ngrp; ++k) gsum = add4(gsum, s_red[k * a.lpr + threadIdx.x]);
    }
    if (PARTS && sg.n_parts > 0) {
        if (base < sg.n_vec) {
            float4 th = ld4(sg.theta + base * 4), m = ld4(sg.m + base * 4), v = ld4(sg.v + base * 4);
            adam4(th, m, v, gsum, lr_t, a.b1, a.b2, a.eps);
            st4(sg.theta + base * 4, th); st4(sg.m + base * 4, m); st4(sg.v + base * 4, v);
        }
        return;
    }
    // Table segment.  All of a thread's loads are issued before anything is consumed (12 float4 + 4 flags in
    // flight per lane): the pass is bound by memory-level parallelism, not by its arithmetic.
    float4 th[kAdamIters], m[kAdamIters], v[kAdamIters];
    int flag[kAdamIters];
    long long vi[kAdamIters];
#pragma unroll
    for (int it = 0; it < kAdamIters; ++it) {
        vi[it] = base + (long long)it * 256;
        flag[it] = 0;
        if (vi[it] < sg.n_vec) {
            flag[it] = sg.touched ? sg.touched[vi[it] >> a.lpr_shift] : (sg.g != nullptr);
            th[it] = ld4(sg.theta + vi[it] * 4); m[it] = ld4(sg.m + vi[it] * 4); v[it] = ld4(sg.v + vi[it] * 4);
        }
    }
    float4 gr[kAdamIters];
#pragma unroll
    for (int it = 0; it < kAdamIters; ++it) {
        gr[it] = make_float4(0, 0, 0, 0);
        if (INDEXED && (flag[it] >> kRefShift)) {
            const int len = flag[it] >> kRefShift;
            const uint32_t pos = (uint32_t)(flag[it] & (int)(kRefMaxRefs - 1));
            const float *src = sg.stage + 4 * (vi[it] & (a.lpr - 1));
            const size_t d = (size_t)4 * a.lpr;
            float4 acc = make_float4(0, 0, 0, 0);
            int k = 0;
            if (sg.sv) {                                         // staging rows in batch order: the list names them
                const uint32_t *sv = sg.sv + pos;
                for (; k + 4 <= len; k += 4) {
                    const uint32_t r0 = sv[k], r1 = sv[k + 1], r2 = sv[k + 2], r3 = sv[k + 3];
                    const float4 x0 = ld4(src + r0 * d), x1 = ld4(src + r1 * d), x2 = ld4(src + r2 * d), x3 = ld4(src + r3 * d);
                    acc = add4(add4(add4(add4(acc, x0), x1), x2), x3);
                }
                for (; k < len; ++k) acc = add4(acc, ld4(src + sv[k] * d));
            } else {                                             // staging rows in LIST order: `len` consecutive rows from `pos`
                const float *run = src + (size_t)pos * d;
                for (; k + 4 <= len; k += 4) {
                    const float4 x0 = ld4(run + (size_t)k * d), x1 = ld4(run + (size_t)(k + 1) * d);
                    const float4 x2 = ld4(run + (size_t)(k + 2) * d), x3 = ld4(run + (size_t)(k + 3) * d);
                    acc = add4(add4(add4(add4(acc, x0), x1), x2), x3);
                }
                for (; k < len; ++k) acc = add4(acc, ld4(run + (size_t)k * d));
            }
            gr[it] = acc;
            if ((vi[it] & (a.lpr - 1)) == 0) sg.touched[vi[it] >> a.lpr_shift] = 0;
        } else if (flag[it]) {               // implies vi < n_vec
            gr[it] = ld4(sg.g + vi[it] * 4);
            if (sg.touched) {                // consume: gradient row and flag back to zero
                st4(sg.g + vi[it] * 4, make_float4(0, 0, 0, 0));
                if ((vi[it] & (a.lpr - 1)) == 0) sg.touched[vi[it] >> a.lpr_shift] = 0;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < kAdamIters; ++it) {
        if (vi[it] < sg.n_vec) {
            adam4(th[it], m[it], v[it], gr[it], lr_t, a.b1, a.b2, a.eps);
            st4(sg.theta + vi[it] * 4, th[it]); st4(sg.m + vi[it] * 4, m[it]); st4(sg.v + vi[it] * 4, v[it]);
        }
    }
}

// deterministic reduction of the step's loss partials (double) by the calling wave -> losses[3]
__device__ __forceinline__ void finalize_losses(const LossArgs &L, int lane, double sq_extra = 0.0) {
    double sq = 0, li = 0, lu = 0, bce = 0, lo = 0;
    for (int k = lane; k < L.n_part; k += 64) {
        const float *o = L.part + (size_t)k * kPartStride;
        sq += o[0]; li += o[1]; lu += o[2]; bce += o[3];
    }
    for (int k = lane; k < L.n_part2; k += 64) sq += L.part2[(size_t)k * kPartStride];
    for (int k = lane; k < L.n_lpart; k += 64) lo += L.lpart[k];
    sq = wave_sum_d(sq) + sq_extra; li = wave_sum_d(li); lu = wave_sum_d(lu); bce = wave_sum_d(bce); lo = wave_sum_d(lo);
    if (lane == 0) {
        const double Bd = (double)L.B;
        float mf;
        if (L.kind == MACR_LOSS_NORMALBCE) {
            mf = (float)(bce / Bd);
        } else {
            const float Lo = (float)(lo / (Bd * Bd)), Li = (float)(li / Bd), Lu = (float)(lu / Bd);
            mf = L.kind == MACR_LOSS_RUBIBCE ? Lo + L.alpha * Li            // macr_mf/model.py:178
                                             : Lo + L.alpha * Li + L.beta * Lu;               // :217
        }
        float regularizer = (float)(0.5 * sq);                  // tf.nn.l2_loss x3  (:219)
        regularizer = regularizer / (float)L.batch_size_cfg;    // (:220)
        const float reg = L.decay * regularizer;                // (:221)
        L.losses[0] = mf + reg; L.losses[1] = mf; L.losses[2] = reg;
    }
}

__global__ __launch_bounds__(64) void k_finalize_losses(LossArgs L) { finalize_losses(L, threadIdx.x); }

// Block 0 also reduces the loss partials of the step into losses[3] (when L.losses is set).
template <bool INDEXED>
__global__ __launch_bounds__(256) void k_adam_dense(AdamArgs a, const StepScalars *scal, LossArgs L) {
    __shared__ float4 s_red[256];
    adam_block<true, INDEXED>(a, blockIdx.x, scal->lr_t, s_red);
    if (blockIdx.x == 0 && L.losses && threadIdx.x < 64) finalize_losses(L, threadIdx.x);
}

// ----------------------------------------------------------------------------
// Lazy dense Adam: the dense pass blocked in TIME (include/macr_hip.h: macr_lazy_adam).
// tf.train.AdamOptimizer moves every row every step (above), and a row the batch did not touch moves by a recurrence that
// needs nothing but the row itself and the step's lr_t:
//     m <- b1 m ;  v <- b2 v ;  theta <- theta - lr_t m / (sqrt(v) + eps)          (adam4 with g = 0, bit for bit *)
// So K such steps can be applied in registers in ONE trip to memory instead of K: every row carries a stamp (the number
// of Adam steps it has received), the library keeps the lr_t of the last kLazyRing steps, and step T
//   * updates the rows the batch touched (stamp .. T-1 without gradient, then step T with it),
//   * sweeps one K-th of the table (chunk c of kAdamVecPerBlock float4 belongs to step T when c % K == T % K) up to T,
//   * reads every other row NOWHERE: 24 d bytes per row every K steps instead of every step.
// Whoever needs a row in between (the gather of the next batch) brings it up to date in registers and writes nothing; a
// flush brings every row to T -- the tables are then what the per-step dense pass leaves, bit for bit (tests: lazy vs dense
// with torch.equal).  The arithmetic per row and step is unchanged, so the pass turns from HBM-bound (33.8 GB per step at
// 11 M rows, d = 128: 6.2 ms) into VALU-bound (13 instructions per element and step: ~0.5 ms).
// (*) with g = 0 the dense pass computes m*b1 + 0 and v*b2 + 0 (as an fma or a mul + add, whichever way the compiler
// contracts them): round(m*b1), round(v*b2) either way; only the sign of a zero can differ, and no later value depends on it.
// ----------------------------------------------------------------------------
constexpr int kLazyRing = 256;
struct LazyState {
    uint32_t t;                  // Adam steps whose lr_t is known: lr[s % kLazyRing] holds step s for s in (t - kLazyRing, t]
    uint32_t pad[3];
    float lr[kLazyRing];
};
static_assert(sizeof(LazyState) == MACR_LAZY_STATE_BYTES, "macr_hip.h: MACR_LAZY_STATE_BYTES");
static_assert(MACR_LAZY_MAX_PERIOD * 2 <= kLazyRing, "a row is at most one sweep period (+ the pending step) behind");

__device__ __forceinline__ void adam4_idle(float4 &th, float4 &m, float4 &v, float lr_t, float b1, float b2, float eps) {
#ifdef MACR_LAZY_IDLE_FULL
    float4 zero = make_float4(0, 0, 0, 0);
    asm volatile("" : "+v"(zero.x), "+v"(zero.y), "+v"(zero.z), "+v"(zero.w));
    adam4(th, m, v, zero, lr_t, b1, b2, eps);
    return;
#endif
    m.x *= b1; m.y *= b1; m.z *= b1; m.w *= b1;
    v.x *= b2; v.y *= b2; v.z *= b2; v.w *= b2;
    th.x = adam_update(th.x, m.x, v.x, lr_t, eps); th.y = adam_update(th.y, m.y, v.y, lr_t, eps);
    th.z = adam_update(th.z, m.z, v.z, lr_t, eps); th.w = adam_update(th.w, m.w, v.w, lr_t, eps);
}

// the step counter advances and the step's lr_t enters the ring (one thread, in a kernel that runs after the step's lr_t
// is known and before the step's lazy pass)
__device__ __forceinline__ void lazy_tick(LazyState *ls, float lr_t) {
    const uint32_t t = ls->t + 1u;
    ls->lr[t & (kLazyRing - 1)] = lr_t;
    ls->t = t;
}

// N float4 of one segment per thread (vi[k] < 0: none): bring each from its row's stamp to step `upto` without gradient,
// then -- STEP -- apply step upto + 1 with the row's gradient (flag protocol of adam_block), store, stamp.  All loads of a
// thread are issued before anything is consumed; the idle steps of the N vectors walk the steps together (one LDS read of
// lr_t per step, N independent chains in flight).
template <int N, bool INDEXED, bool STEP>
__device__ __forceinline__ void adam_lazy_vecs(const AdamArgs &a, const AdamSeg &sg, const long long (&vi)[N], uint32_t upto,
                                               const float *s_lr) {
    float4 th[N], m[N], v[N], gr[N];
    int flag[N];
    uint32_t stamp[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        flag[k] = 0; stamp[k] = upto;
        if (vi[k] >= 0) {
            const long long row = vi[k] >> a.lpr_shift;
            if (STEP) flag[k] = sg.touched[row];
            stamp[k] = sg.stamp[row];
            th[k] = ld4(sg.theta + vi[k] * 4); m[k] = ld4(sg.m + vi[k] * 4); v[k] = ld4(sg.v + vi[k] * 4);
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        gr[k] = make_float4(0, 0, 0, 0);
        if (!STEP) continue;
        if (INDEXED && (flag[k] >> kRefShift)) {                // (see adam_block: the row's staged gradient rows, in list order)
            const int len = flag[k] >> kRefShift;
            const uint32_t pos = (uint32_t)(flag[k] & (int)(kRefMaxRefs - 1));
            const float *src = sg.stage + 4 * (vi[k] & (a.lpr - 1));
            const size_t d = (size_t)4 * a.lpr;
            float4 acc = make_float4(0, 0, 0, 0);
            int q = 0;
            if (sg.sv) {
                const uint32_t *sv = sg.sv + pos;
                for (; q + 4 <= len; q += 4) {
                    const uint32_t r0 = sv[q], r1 = sv[q + 1], r2 = sv[q + 2], r3 = sv[q + 3];
                    const float4 x0 = ld4(src + r0 * d), x1 = ld4(src + r1 * d), x2 = ld4(src + r2 * d), x3 = ld4(src + r3 * d);
                    acc = add4(add4(add4(add4(acc, x0), x1), x2), x3);
                }
                for (; q < len; ++q) acc = add4(acc, ld4(src + sv[q] * d));
            } else {
                const float *run = src + (size_t)pos * d;
                for (; q + 4 <= len; q += 4) {
                    const float4 x0 = ld4(run + (size_t)q * d), x1 = ld4(run + (size_t)(q + 1) * d);
                    const float4 x2 = ld4(run + (size_t)(q + 2) * d), x3 = ld4(run + (size_t)(q + 3) * d);
                    acc = add4(add4(add4(add4(acc, x0), x1), x2), x3);
                }
                for (; q < len; ++q) acc = add4(acc, ld4(run + (size_t)q * d));
            }
            gr[k] = acc;
        } else if (flag[k] & 1) {                               // (bit 1: adam_lazy_scan_block)
            gr[k] = ld4(sg.g + vi[k] * 4);
            st4(sg.g + vi[k] * 4, make_float4(0, 0, 0, 0));
        }
    }
    uint32_t lag = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) lag = upto - stamp[k] > lag ? upto - stamp[k] : lag;       // (stamp <= upto always)
    for (uint32_t s = upto - lag + 1u; s - 1u != upto; ++s) {                                 // s = upto - lag + 1 .. upto
        const float lr_s = s_lr[s & (kLazyRing - 1)];
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (s - 1u - stamp[k] < lag) adam4_idle(th[k], m[k], v[k], lr_s, a.b1, a.b2, a.eps);   // stamp[k] < s (unsigned: s - 1 - stamp in [0, lag))
    }
    const float lr_T = s_lr[(upto + 1u) & (kLazyRing - 1)];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (vi[k] < 0) continue;
        if (STEP) adam4(th[k], m[k], v[k], gr[k], lr_T, a.b1, a.b2, a.eps);
        st4(sg.theta + vi[k] * 4, th[k]); st4(sg.m + vi[k] * 4, m[k]); st4(sg.v + vi[k] * 4, v[k]);
        if ((vi[k] & (a.lpr - 1)) == 0) {
            const long long row = vi[k] >> a.lpr_shift;
            sg.stamp[row] = STEP ? upto + 1u : upto;
            if (STEP && flag[k]) sg.touched[row] = 0;
        }
    }
}

// the lazy pass of one step (STEP) or the flush (!STEP).  Grid: sweep blocks of every table segment (n_sweep[s] each, block k
// of a segment = chunk phase + k * period), then -- STEP -- n_touch blocks over the sorted reference list (one LPR-lane group per
// position; the group at a row's first reference updates the row unless this step's sweep does), then one block per
// branch-vector segment (dense every step: one row each).
struct LazyArgs {
    const LazyState *state;
    int period;
    long long sweep_first[3];        // first block of segment s's sweep blocks; [n_tab] = their end
    int n_tab;                       // table segments = a.seg[0 .. n_tab)
    long long touch_first, branch_first;
    int n_refs, n_users, item_seg;   // sorted reference list: keys [0, n_users) = user rows (segment 0), [n_users, key_end) = item rows
    uint32_t key_end;
    const uint32_t *sk;
};

template <bool INDEXED, bool STEP>
__global__ __launch_bounds__(256) void k_adam_lazy(AdamArgs a, LazyArgs z, const StepScalars *scal) {
    __shared__ float s_lr[kLazyRing];
    __shared__ float4 s_red[256];
    const long long blk = blockIdx.x;
    if (STEP && blk >= z.branch_first) {
        adam_block<true, false>(a, a.seg[z.n_tab + (int)(blk - z.branch_first)].first_block, scal->lr_t, s_red);
        return;
    }
    s_lr[threadIdx.x] = z.state->lr[threadIdx.x];
    const uint32_t T = z.state->t;                  // STEP: the step being applied; flush: the last step applied anywhere
    const uint32_t upto = STEP ? T - 1u : T;
    const uint32_t phase = T % (uint32_t)z.period;
    __syncthreads();
    if (blk < z.touch_first) {
        int s = 0;
        if (z.n_tab > 1 && blk >= z.sweep_first[1]) s = 1;
        const AdamSeg &sg = a.seg[s];
        const long long chunk = (STEP ? (long long)phase : 0) + (blk - z.sweep_first[s]) * (STEP ? z.period : 1);
        const long long base = chunk * kAdamVecPerBlock + threadIdx.x;
        long long vi[kAdamIters];
#pragma unroll
        for (int it = 0; it < kAdamIters; ++it) {
            vi[it] = base + (long long)it * 256;
            if (vi[it] >= sg.n_vec) vi[it] = -1;
        }
        adam_lazy_vecs<kAdamIters, INDEXED, STEP>(a, sg, vi, upto, s_lr);
        return;
    }
    if (STEP) {
        const int sub = threadIdx.x & (a.lpr - 1);
        const long long p = (blk - z.touch_first) * (256 >> a.lpr_shift) + (threadIdx.x >> a.lpr_shift);
        long long vi[1] = {-1};
        int s = 0;
        if (p < z.n_refs) {
            const uint32_t key = z.sk[p], prev = p > 0 ? z.sk[p - 1] : 0xffffffffu;
            if (key < z.key_end && key != prev) {
                const bool is_user = key < (uint32_t)z.n_users;
                s = is_user ? 0 : z.item_seg;
                const long long row = is_user ? key : key - (uint32_t)z.n_users;
                const long long chunk = (row << a.lpr_shift) / kAdamVecPerBlock;      // (a row never straddles two chunks)
                if (chunk % z.period != phase) vi[0] = (row << a.lpr_shift) + sub;
            }
        }
        if (vi[0] >= 0) adam_lazy_vecs<1, INDEXED, true>(a, a.seg[s], vi, upto, s_lr);
    }
}

// The lazy pass of the SMALL tables (macr_mf_train_step in deferred mode): no sorted reference list names the touched rows,
// but scanning every row's flag is a few hundred KB, so block `blk` owns chunk blk of its segment and updates
//   - every row of the chunk when the chunk belongs to this step's sweep,
//   - else the rows with a flag: bit 0 = pair_bwd left a gradient row for the pending step, bit 1 = pair_fwd<2> marked
//     the row as one of the CURRENT batch (pair_bwd of this step reads its rows in place, so they must be at the step).
// Per step that is the rows of two batches + a K-th of the tables instead of every row.  Branch-vector segments stay
// dense (one row each).  s_red doubles as the block's copy of the lr_t ring.
__device__ __forceinline__ void adam_lazy_scan_block(const AdamArgs &a, long long blk, float lr_t, float4 *s_red) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.n_seg && blk >= a.seg[k].first_block) s = k;
    if (a.seg[s].n_parts > 0) { adam_block<true, false>(a, blk, lr_t, s_red); return; }
    float *s_lr = reinterpret_cast<float *>(s_red);
    s_lr[threadIdx.x] = a.lazy->lr[threadIdx.x];
    const uint32_t T = a.lazy->t;                   // the pending step
    __syncthreads();
    const AdamSeg &sg = a.seg[s];
    const long long chunk = blk - sg.first_block;
    const bool sweep = (uint32_t)(chunk % a.period) == T % (uint32_t)a.period;
    long long vi[kAdamIters];
#pragma unroll
    for (int it = 0; it < kAdamIters; ++it) {
        const long long v = chunk * kAdamVecPerBlock + threadIdx.x + (long long)it * 256;
        vi[it] = -1;
        if (v < sg.n_vec && (sweep || sg.touched[v >> a.lpr_shift] != 0)) vi[it] = v;
    }
    adam_lazy_vecs<kAdamIters, false, true>(a, sg, vi, T - 1u, s_lr);
}

// the same pass as a launch of its own (a step that completes in its call, macr_mf_train_flush); block 0 also reduces the loss partials
__global__ __launch_bounds__(256) void k_adam_lazy_scan(AdamArgs a, const StepScalars *scal, LossArgs L) {
    __shared__ float4 s_red[256];
    adam_lazy_scan_block(a, blockIdx.x, scal->lr_t, s_red);
    if (blockIdx.x == 0 && L.losses && threadIdx.x < 64) finalize_losses(L, threadIdx.x);
}

// rows of a lazily updated table as the per-step dense pass would hold them now: theta brought from the row's stamp to the
// current step in registers, nothing written back.  One LPR-lane group per output row; rows[k] < 0: a zero row.
struct LazyTable { const float *theta, *m, *v; const uint32_t *stamp; };
template <int LPR>
__device__ __forceinline__ float4 lazy_row4(const LazyTable &tb, long long row, int sub, uint32_t T, const float *s_lr, float b1,
                                            float b2, float eps) {
    const size_t at = ((size_t)row * LPR + sub) * 4;
    float4 th = ld4(tb.theta + at);
    if (tb.stamp) {
        const uint32_t st = tb.stamp[row];
        if (st != T) {
            float4 m = ld4(tb.m + at), v = ld4(tb.v + at);
            for (uint32_t s = st + 1u; s - 1u != T; ++s) adam4_idle(th, m, v, s_lr[s & (kLazyRing - 1)], b1, b2, eps);
        }
    }
    return th;
}
template <int LPR>
__global__ __launch_bounds__(256) void k_lazy_rows(long long n, const int32_t *__restrict__ rows, LazyTable tb, const LazyState *state,
                                                   float b1, float b2, float eps, float *__restrict__ out) {
    __shared__ float s_lr[kLazyRing];
    s_lr[threadIdx.x] = state->lr[threadIdx.x];
    const uint32_t T = state->t;
    __syncthreads();
    const int sub = threadIdx.x % LPR;
    for (long long k = blockIdx.x * (256LL / LPR) + threadIdx.x / LPR; k < n; k += gridDim.x * (256LL / LPR)) {
        const int row = rows[k];
        float4 x = make_float4(0, 0, 0, 0);
        if (row >= 0) x = lazy_row4<LPR>(tb, row, sub, T, s_lr, b1, b2, eps);
        st4(out + ((size_t)k * LPR + sub) * 4, x);
    }
}

// ----------------------------------------------------------------------------
// pair_fwd: gathers + per-pair dots + branch logits.
//   eu=Usrc[u], ei=Isrc[i], ej=Isrc[j]                    macr_mf/model.py:35-37
//   p=sum(eu*ei), n=sum(eu*ej)                            :186-187
//   si=ei.w, sj=ej.w, su=eu.wu                            :194-196
//   a=sig(si)sig(su), b=sig(sj)sig(su)  (row factors of the (B,B) products :204-205)
//   partials: l2 regulariser sum (:219), L_item (:213), L_user (:215) terms
// fwd layout: 7 arrays of Bp floats: p, n, a, b, sig_si, sig_sj, sig_su  (Bp = padded B).
// ----------------------------------------------------------------------------
// Window of the (B,B) kernel's 4-transcendental form in y = n*b (see k_bxb): shared with pair_fwd, which flags the columns outside it
constexpr float kBxbYHi = 6.0f, kBxbYLoPairs = -60.0f, kBxbYLoSingle = -20.0f;
constexpr int kNeutralTileMax = 8;      // a 64-column tile with more flagged columns than this is not neutralised (its wave decides as before)

// Tables the deferred-mode forward needs to see one update ahead (below).
struct PendingAdam {
    const float *mU, *vU, *gU, *mI, *vI, *gI;      // slots and gradient sums of the user / item table
    int32_t *tU, *tI;                                // row flags: gradient present (PENDING = 2: bit 1 added for the batch's rows)
    const float *mw, *vw, *mwu, *vwu;                // slots of the branch vectors
    const StepScalars *scal;
    float b1, b2, eps;
    const uint32_t *stU, *stI;                       // PENDING = 2 (lazy dense Adam): row stamps, step counter + lr_t ring
    const LazyState *lazy;
};

// PENDING (deferred mode): the previous step's Adam update has not been applied yet -- it is applied to ALL rows
// by the Adam blocks riding in this step's bxb launch.  The forward needs the updated rows now, so every lane
// computes theta' = adam(theta, m, v, g) for the float4 it gathers (and for w, w_user) in registers, with exactly
// the arithmetic the pass will use, and writes nothing back: 3x more bytes gathered per row, no atomics, no
// ordering between references to the same row, no extra launch.
// PENDING = 2: the tables are updated lazily (adam_lazy_scan_block) -- a gathered row may be several steps behind: its idle
// steps first (stamp + 1 .. pending step - 1, in registers), then the pending step as above.  The lane group also marks
// its three rows (flag bit 1) so that the pass riding in this step's (B,B) launch brings them to the pending step IN
// MEMORY before pair_bwd reads them in place.
template <int LPR, int PENDING>
__global__ __launch_bounds__(256) void k_pair_fwd(
    int B, int Bp, const int32_t *__restrict__ u, const int32_t *__restrict__ i, const int32_t *__restrict__ j,
    const float *__restrict__ Usrc, const float *__restrict__ Isrc,
    const float *__restrict__ w, const float *__restrict__ wu,
    float *__restrict__ fwd, float *__restrict__ part, int reg_on_gathered, float *__restrict__ gw, PendingAdam pa,
    int user_branch) {
    constexpr int d = 4 * LPR;
    __shared__ float red[48];
    __shared__ float s_lr[PENDING == 2 ? kLazyRing : 1];
    RowGroup<LPR> g;
    const int t = blockIdx.x * RowGroup<LPR>::kRowsPerBlock + g.slot;
    float sq = 0.f, litem = 0.f, luser = 0.f;
    uint32_t T = 0;
    if (PENDING == 2) {
        s_lr[threadIdx.x] = pa.lazy->lr[threadIdx.x];
        T = pa.lazy->t;
        __syncthreads();
    }
    if (!PENDING && blockIdx.x == 0) {
        // nothing reads the branch-vector partial rows before this step's pair_bwd adds into them
        for (int k = threadIdx.x; k < kBranchSlots * 2 * d; k += 256) gw[k] = 0.f;
    }
    if (t < B) {
        const int ru = u[t], ri = i[t], rj = j[t];
        float4 eu = ld4(Usrc + (size_t)ru * d + 4 * g.sub);
        float4 ei = ld4(Isrc + (size_t)ri * d + 4 * g.sub);
        float4 ej = ld4(Isrc + (size_t)rj * d + 4 * g.sub);
        float4 w4 = ld4(w + 4 * g.sub), wu4 = ld4(wu + 4 * g.sub);
        if (PENDING) {
            const float lr_t = pa.scal->lr_t;
            const size_t au = (size_t)ru * d + 4 * g.sub, ai = (size_t)ri * d + 4 * g.sub, aj = (size_t)rj * d + 4 * g.sub;
            float4 mu = ld4(pa.mU + au), vu = ld4(pa.vU + au), mi = ld4(pa.mI + ai), vi = ld4(pa.vI + ai);
            float4 mj = ld4(pa.mI + aj), vj = ld4(pa.vI + aj);
            // gradient rows are read unconditionally: rows without a gradient are all-zero (the scratch invariant),
            // and a flag test would put a second memory latency in front of the row load
            const float4 gu = ld4(pa.gU + au), gi = ld4(pa.gI + ai), gj = ld4(pa.gI + aj);
            float4 gw4 = make_float4(0, 0, 0, 0), gwu4 = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < kBranchSlots; ++k) {
                gw4 = add4(gw4, ld4(gw + (size_t)k * 2 * d + 4 * g.sub));
                gwu4 = add4(gwu4, ld4(gw + (size_t)k * 2 * d + d + 4 * g.sub));
            }
            float4 mw4 = ld4(pa.mw + 4 * g.sub), vw4 = ld4(pa.vw + 4 * g.sub);
            float4 mwu4 = ld4(pa.mwu + 4 * g.sub), vwu4 = ld4(pa.vwu + 4 * g.sub);
            if (PENDING == 2) {
                const uint32_t upto = T - 1u, su_ = pa.stU[ru], si_ = pa.stI[ri], sj_ = pa.stI[rj];
                if (g.sub == 0) {                        // (every writer of a flag writes old | 2: equal rows of the batch agree)
                    const int fu = pa.tU[ru], fi = pa.tI[ri], fj = pa.tI[rj];
                    if (!(fu & 2)) pa.tU[ru] = fu | 2;
                    if (!(fi & 2)) pa.tI[ri] = fi | 2;
                    if (!(fj & 2)) pa.tI[rj] = fj | 2;
                }
                uint32_t lag = upto - su_;
                lag = upto - si_ > lag ? upto - si_ : lag;
                lag = upto - sj_ > lag ? upto - sj_ : lag;
                for (uint32_t s = upto - lag + 1u; s - 1u != upto; ++s) {
                    const float lr_s = s_lr[s & (kLazyRing - 1)];
                    if (s - 1u - su_ < lag) adam4_idle(eu, mu, vu, lr_s, pa.b1, pa.b2, pa.eps);
                    if (s - 1u - si_ < lag) adam4_idle(ei, mi, vi, lr_s, pa.b1, pa.b2, pa.eps);
                    if (s - 1u - sj_ < lag) adam4_idle(ej, mj, vj, lr_s, pa.b1, pa.b2, pa.eps);
                }
            }
            adam4(eu, mu, vu, gu, lr_t, pa.b1, pa.b2, pa.eps);
            adam4(ei, mi, vi, gi, lr_t, pa.b1, pa.b2, pa.eps);
            adam4(ej, mj, vj, gj, lr_t, pa.b1, pa.b2, pa.eps);
            adam4(w4, mw4, vw4, gw4, lr_t, pa.b1, pa.b2, pa.eps);
            adam4(wu4, mwu4, vwu4, gwu4, lr_t, pa.b1, pa.b2, pa.eps);
        }
        const float p = group_sum<LPR>(dot4(eu, ei));
        const float n = group_sum<LPR>(dot4(eu, ej));
        const float si = group_sum<LPR>(dot4(ei, w4));
        const float sj = group_sum<LPR>(dot4(ej, w4));
        const float su = group_sum<LPR>(dot4(eu, wu4));
        if (reg_on_gathered) sq = dot4(eu, eu) + dot4(ei, ei) + dot4(ej, ej);
        if (g.sub == 0) {
            const float eps = 1e-10f;
            // MACR_LOSS_RUBIBCE (model.py:158-183) is this graph without the user factor: sig(su) := 1, which also
            // zeroes d/dsu in pair_bwd (ssu*(1-ssu) = 0 and both L_user derivatives vanish at 1); L_user is not in the loss
            const float ssi = sigmoid_acc(si), ssj = sigmoid_acc(sj), ssu = user_branch ? sigmoid_acc(su) : 1.0f;
            fwd[0 * (size_t)Bp + t] = p;
            fwd[1 * (size_t)Bp + t] = n;
            fwd[2 * (size_t)Bp + t] = ssi * ssu;
            fwd[3 * (size_t)Bp + t] = ssj * ssu;
            fwd[4 * (size_t)Bp + t] = ssi;
            fwd[5 * (size_t)Bp + t] = ssj;
            // The SIGN of the stored sig(su) (a value in [0, 1]; its readers take fabsf) flags column t of the (B,B) term as outside
            // the window of the 4-transcendental form whatever its row (a, b <= 1): k_bxb takes such columns out of the rotation
            // (see "neutralised columns" there).  An array of its own cost pair_fwd 0.5 us for the extra store.
            fwd[6 * (size_t)Bp + t] = (p < -20.0f || n > kBxbYHi || n < kBxbYLoPairs) ? -ssu : ssu;
            litem = -logf(ssi + eps) + -logf((1.0f - ssj) + eps);
            luser = user_branch ? -logf(ssu + eps) + -logf((1.0f - ssu) + eps) : 0.0f;
        }
    }
    float s0 = sq, s1 = litem, s2 = luser;
    block_sum3(s0, s1, s2, red);                     // (one pair of barriers for the three sums)
    if (threadIdx.x == 0) {
        float *o = part + (size_t)blockIdx.x * kPartStride;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = 0.f;
    }
}

// ----------------------------------------------------------------------------
// bxb: the (B,B) broadcast term of rubibceboth, entirely on chip.
//   X[r,c]=a[r]*p[c], Y[r,c]=b[r]*n[c]                       macr_mf/model.py:204-205
//   L_ori = mean(-log(sig(X)+1e-10) - log(1-sig(Y)+1e-10))   :211
// and its gradient: row sums da,db and column sums dp,dn of f'(X), g'(Y).
//
// A wave owns a (64*R)-row x 64-column tile: lane t holds rows t, t+64, ... (R of them) for
// the whole tile and, at iteration k, column (t+k) mod 64.  The column's inputs (p,n) and
// its running column sums travel with it: after every iteration the four registers are
// rotated one lane down the wave with DPP (v_mov_b32_dpp wave_rol:1), so after 64
// iterations each column sum is back in its home lane holding the total over the wave's
// rows; R rows per lane amortise the four rotations over R pairs.  Row sums never leave
// their lane.  No LDS traffic, no atomics in the inner loop.
// The X and Y halves of a pair are the two lanes of packed fp32 math (v_pk_mul/add/fma_f32:
// two results per lane per issue), which leaves the kernel bound by its transcendentals
// (~8.5 cycles per wave64 each): 4 per pair on the FAST path, 6 on the EXACT path (see
// run_tile below; the reference writes 8: 2 exp, 2+2 rcp, 2 log).
// Block = 4 waves = (64*R) rows x 256 columns (one 64-column tile per wave); the waves' row
// sums meet in LDS once.   rowpart [ncb][2][Bp], colpart [nrb][2][Bp], lpart [nrb*ncb]
// ----------------------------------------------------------------------------
// ----------------------------------------------------------------------------
// Batch grouping (small batches).  The order of the triples of a batch is free -- every loss term is a sum over the
// batch -- and the reference's sampler draws positives by popularity (macr_mf/load_data.py:543-566): a batch
// references a few item rows hundreds of times, and hundreds of atomics on one L2 line serialise (~24 ns each).
// The batch is therefore bucketed by the low byte of the positive item -- a single stable counting pass: one trip to
// global memory, 4 KB of LDS per workgroup -- so that equal positive items land in the same 16-slot chunks of
// pair_bwd, which adds equal rows of a chunk once (combine_positive_rows).  The workgroups are extra blocks of the
// (B,B) launch (rubibceboth: 14+ us of cover) or a tiny launch of their own (normalbce).
// Output: the batch in bucket order (us, is, js) and perm[s] = position in the caller's batch of the triple at slot s
// (its index in fwd and in the bxb partials).  Measured before settling on this: a full 16-bit sort by one workgroup
// through global ping-pong buffers took 40 us (two trips per pass and phase), and its LDS version would have taken
// the LDS of the (B,B) blocks it rides with.
// ----------------------------------------------------------------------------
struct BatchSort {
    const int32_t *u, *i, *j;
    int B;
    int rb0;                            // (B,B) launch: first row block of the launch (row-sharded training), else 0
    int32_t *perm, *us, *is, *js;       // [B] each
    // (B,B) blocks: neutralised columns (k_bxb).  outflag = pair_fwd's column flags = the SIGNS of fwd[6] (NULL: off), nneu = neutral blocks per row block,
    // ncbx = ncb + nneu = row-sum slabs / loss partials per row block
    const float *outflag;
    int nneu, ncbx;
};

constexpr int kBucketSpan = 512;        // triples bucketed by one workgroup (independently of the other spans)

// Workgroup `part` buckets slots [part*kBucketSpan, (part+1)*kBucketSpan) of the batch among themselves: a hot item
// is spread evenly over the batch, so grouping inside spans of 512 combines as well as grouping the whole batch,
// and B/512 workgroups finish in a quarter of the time one needs for B = 4096 (the pass is a chain of LDS and
// global-memory latencies, not throughput).
template <int NW>
__device__ __forceinline__ void batch_bucket_block(const BatchSort &s, int part, uint32_t *s_hist /* NW*256 + 4 words */) {
    const int t = threadIdx.x, wid = t >> 6, lane = t & 63;
    const int plo = part * kBucketSpan, phi = plo + kBucketSpan < s.B ? plo + kBucketSpan : s.B;
    constexpr int per = kBucketSpan / NW;
    const int lo = plo + wid * per < phi ? plo + wid * per : phi, hi = lo + per < phi ? lo + per : phi;
    const uint32_t *keys = reinterpret_cast<const uint32_t *>(s.i);
    for (int k = t; k < NW * kRadix; k += NW * 64) s_hist[k] = 0;
    __syncthreads();
    wave_count(keys, lo, hi, 0, s_hist + wid * kRadix);
    __syncthreads();
    block_digit_offsets<NW>(s_hist, s_hist + NW * kRadix, nullptr, 0);
    uint32_t *offs = s_hist + wid * kRadix;
    const uint64_t below = (1ull << lane) - 1ull;
    constexpr int NQ = per / 64;
    int32_t ri[NQ], ru[NQ], rj[NQ];
    if (lo < hi) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                                        // clamped, unpredicated: all in flight at once
            const int p = lo + q * 64 + lane, pc = p < hi ? p : hi - 1;
            ri[q] = s.i[pc]; ru[q] = s.u[pc]; rj[q] = s.j[pc];
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (lo + q * 64 >= hi) break;                                     // wave-uniform
        const int p = lo + q * 64 + lane;
        const bool valid = p < hi;
        const uint32_t dgt = (uint32_t)ri[q] & 255u;
        const uint64_t peers = match_digit(dgt, valid);
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        const uint32_t dst = plo + offs[dgt] + rank;
        if (valid && rank == 0) offs[dgt] = dst - plo + (uint32_t)__popcll(peers);
        if (valid) { s.perm[dst] = p; s.us[dst] = ru[q]; s.is[dst] = ri[q]; s.js[dst] = rj[q]; }
    }
}

__global__ __launch_bounds__(256) void k_batch_sort(BatchSort s) {
    __shared__ uint32_t s_hist[4 * kRadix + 4];
    batch_bucket_block<4>(s, blockIdx.x, s_hist);
}

__device__ __forceinline__ float wave_rol1(float v) {      // lane i <- lane (i+1) mod 64
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x134, 0xf, 0xf, true));
}

typedef float v2f __attribute__((ext_vector_type(2)));

// ADAM (deferred mode): blocks [nbxb, gridDim.x) of the launch are Adam blocks completing the PREVIOUS step's
// dense pass (they touch only theta/m/v/g, never the bxb inputs): the bxb blocks are resident for the whole
// launch and bound by the transcendental rate, the Adam blocks stream through the remaining wave slots and are
// bound by HBM, so the two costs overlap instead of adding.
// ADAM = 2: the lazy pass (adam_lazy_scan_block) in place of the dense one.
// ADAM 0 / 1 are compiled for six waves per SIMD (67 / 80 VGPRs, no spills; the compiler's own choice is 94 = five waves): the dense Adam
// blocks riding in the launch are bound by memory-level parallelism, and one more of their waves beside each (B,B) wave is worth
// 0.3-0.5 us per step -- bxb+adam 19.12-19.41 -> 18.87-19.17 us, bxb alone 12.7-13.2 -> 12.1-12.6, same box (profiles/
// r06_bxb_waves_ab.txt; seven waves spill and cost 4.5 us).  The lazy form (ADAM == 2) needs its registers: left alone.
template <int R, bool FULL, int ADAM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ADAM == 2 ? 4 : 6, 8))) void k_bxb(int B, int Bp, int ncb, int nbxb, const float *__restrict__ fwd,
                                             float *__restrict__ rowpart, float *__restrict__ colpart,
                                             float *__restrict__ lpart, AdamArgs adam, const StepScalars *scal,
                                             BatchSort sort) {
    constexpr int RB = 64 * R;                 // rows per block
    __shared__ float s_row[4][2][RB];
    __shared__ float red[16];
    __shared__ float4 s_red[ADAM ? 256 : 1];
    __shared__ uint32_t s_hist[4 * kRadix + 4];
    const int nsort = (sort.B + kBucketSpan - 1) / kBucketSpan;
    const int ncbx = sort.nneu ? sort.ncbx : ncb;      // row-sum slabs / loss partials per row block
    const int nbn = sort.nneu * (nbxb / ncb);          // neutral blocks (below): nneu per row block of the launch
    if ((int)blockIdx.x >= nbxb && (int)blockIdx.x < nbxb + nbn) {
        // NEUTRAL BLOCK (rb, part): the exact form for the columns the rotation blocks took out (see there), against the rows of
        // row block rb; flagged column number idx (in column order, over the neutralised tiles) belongs to wave
        // idx mod (4 nneu) of the row block's nneu blocks.  Row sums and loss go to slab / partial ncb + part.
        const int nb_idx = (int)blockIdx.x - nbxb, part = nb_idx % sort.nneu, rb = sort.rb0 + nb_idx / sort.nneu;
        const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
        const float *p = fwd, *n = fwd + Bp, *a = fwd + 2 * (size_t)Bp, *b = fwd + 3 * (size_t)Bp;
        const float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
        const v2f one = {1.0f, 1.0f}, eps = {1e-10f, 1e-10f};
        v2f abn[R], dabn[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int r = rb * RB + q * 64 + lane;
            abn[q] = v2f{a[r], b[r]};
            dabn[q] = v2f{0.f, 0.f};
        }
        float l2n = 0.f;
        const int slot = part * 4 + wid, nslot = 4 * sort.nneu;
        int idx = 0;
        for (int base = 0; base < B; base += 1024) {            // 1024 columns per trip: lane l holds columns base + 16 l .. + 15 (4 x float4, all in flight)
            const int at = base + 16 * lane;
            uint32_t bits = 0;                                  // bit k: column at + k flagged
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 f4 = at < B ? ld4(sort.outflag + at + 4 * e) : make_float4(0, 0, 0, 0);
                bits |= ((__float_as_uint(f4.x) >> 31) | ((__float_as_uint(f4.y) >> 31) << 1) | ((__float_as_uint(f4.z) >> 31) << 2) |
                         ((__float_as_uint(f4.w) >> 31) << 3)) << (4 * e);
            }
            int mine = __popc(bits);
            const int pair2 = mine + __shfl_xor(mine, 1, 64);
            const int tile = pair2 + __shfl_xor(pair2, 2, 64);             // sum over the 4 lanes of a 64-column tile
            if (tile > kNeutralTileMax) mine = 0;               // (that tile's waves decide for themselves)
            for (uint64_t hit = __ballot(mine != 0); hit; hit &= hit - 1) {
                const int src = __builtin_ctzll(hit);
                for (uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)bits, src); wv; wv &= wv - 1) {
                    const int mine_now = idx++ % nslot == slot;
                    if (!mine_now) continue;                    // (wave-uniform)
                    const int cj = base + 16 * src + __builtin_ctz(wv);
                    const v2f pn = {p[cj], n[cj]};
                    v2f accj = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < R; ++q) {
                        const v2f z = pn * (abn[q] * (-kLog2e));
                        const v2f dd = v2f{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + one;
                        const v2f sg = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};
                        const v2f om = one - sg;
                        const v2f ts = sg + eps, tom = om + eps;
                        const float txy = ts.x * tom.y;
                        l2n += __builtin_amdgcn_logf(txy);
                        const float r2 = __builtin_amdgcn_rcpf(txy);
                        const v2f h = (sg * om) * r2;
                        const v2f g = v2f{h.x * tom.y, h.y * ts.x};
                        dabn[q] = __builtin_elementwise_fma(g, pn, dabn[q]);
                        accj = __builtin_elementwise_fma(g, abn[q], accj);
                    }
                    const float sx = wave_sum(accj.x), sy = wave_sum(accj.y);
                    if (lane == 0) {
                        colpart[((size_t)rb * 2 + 0) * Bp + cj] = -sx;
                        colpart[((size_t)rb * 2 + 1) * Bp + cj] = sy;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            s_row[wid][0][q * 64 + lane] = -dabn[q].x;
            s_row[wid][1][q * 64 + lane] = dabn[q].y;
        }
        const float lsum = block_sum(-l2n * kLn2, red);
        if (t == 0) lpart[(size_t)rb * ncbx + ncb + part] = lsum;
        for (int e = t; e < 2 * RB; e += 256) {
            const int q = e / RB, rr = e % RB, r = rb * RB + rr;
            rowpart[((size_t)(ncb + part) * 2 + q) * Bp + r] = (s_row[0][q][rr] + s_row[1][q][rr]) + (s_row[2][q][rr] + s_row[3][q][rr]);
        }
        return;
    }
    if ((int)blockIdx.x >= nbxb + nbn && (int)blockIdx.x < nbxb + nbn + nsort) {     // these blocks group the batch for pair_bwd
        __builtin_amdgcn_s_setprio(3);         // a chain of latencies beside VALU-bound waves: go first when ready
#ifndef MACR_ABL_NOGROUP
        batch_bucket_block<4>(sort, blockIdx.x - nbxb - nbn, s_hist);
#endif
        return;
    }
    const bool is_adam = ADAM && (int)blockIdx.x >= nbxb + nbn + nsort;
    const int ablk = blockIdx.x - nbxb - nbn - nsort, bblk = blockIdx.x;
    if (is_adam) {
        // the bxb waves are older and would win every issue slot: the Adam waves (a handful of VALU instructions
        // between long memory waits) go first whenever they are ready
#ifndef MACR_ABL_ADAM_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        if (ADAM == 2) adam_lazy_scan_block(adam, (long long)ablk, scal->lr_t, s_red);
        else adam_block<true>(adam, (long long)ablk, scal->lr_t, s_red);
        return;
    }
    const int cb = bblk % ncb, rb = sort.rb0 + bblk / ncb, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const float *p = fwd, *n = fwd + Bp, *a = fwd + 2 * (size_t)Bp, *b = fwd + 3 * (size_t)Bp;
    const float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    const v2f one = {1.0f, 1.0f}, eps = {1e-10f, 1e-10f};
    v2f ab[R], abs_[R], dab[R];                // {a,b} of the lane's rows; the same scaled by -log2(e); row sums
    bool rok[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int r = rb * RB + q * 64 + lane;
        rok[q] = FULL || r < B;
        ab[q] = v2f{rok[q] ? a[r] : 0.f, rok[q] ? b[r] : 0.f};
        abs_[q] = ab[q] * (-kLog2e);
        dab[q] = v2f{0.f, 0.f};
    }
    const int c = cb * 256 + wid * 64 + lane;  // the column this lane is home to
    const bool cok = FULL || c < B;
    v2f cn = {cok ? p[c] : 0.f, cok ? n[c] : 0.f};
    // NEUTRALISED COLUMNS (sort.nneu: full batches).  One column outside the window of the 4-transcendental form (below) sends its
    // whole 64-column tile through the exact form (1.24x the time); a model a few hundred steps old has 0.1-1 % of them
    // (negatives scoring above 6), almost half of its tiles hold one -- and at one wave per SIMD the launch lasts as long as
    // its slowest wave.  So a column that pair_fwd flagged (outside the window for ANY row) rotates as (p, n) = (0, 0): its
    // contributions to the row sums are g * 0 = 0 exactly, its own column sums are dropped, and each of its evaluations adds
    // exactly log2(1/2 * 1/2) = -2 to the loss sum, which is taken back below.  The flagged columns are evaluated by the exact
    // form in the launch's NEUTRAL BLOCKS (above): a few (column, R rows) evaluations per wave there instead of 64 R exact
    // ones here.  A tile with more than kNeutralTileMax flagged columns stays as it is (a saturated model: no cliff).
    bool own_out = false;
    if (FULL && sort.nneu) {
        own_out = (__float_as_uint(sort.outflag[c]) >> 31) != 0u;
        const int lw = (int)__popcll(__ballot(own_out));
        if (lw > kNeutralTileMax) own_out = false;
        if (own_out) cn = v2f{0.f, 0.f};
    }
    v2f acc = {0.f, 0.f}, l2 = {0.f, 0.f};     // column sums travelling with cn; log2 terms (scaled by ln2 at the end)
    // Two forms of the pair arithmetic.  EXACT follows the reference operation by operation: s = 1/(1+e^-z), then
    // s+eps and (1-s)+eps with 1-s by subtraction (6 transcendentals with the shared log/rcp of tx*ty).  FAST works
    // on numerators and denominators, with ex = e^-x, dx = 1+ex, nx = 1+eps*dx (so s+eps = nx/dx) and ey, dy,
    // ny = ey+eps*dy (so (1-s)+eps = ny/dy):
    //     log(tx*ty) = log(N^2 * r),  f'(x) = -ex*r*(dy*ny),  g'(y) = ey*r*(dx*nx),   N = nx*ny,  r = 1/(dx*nx*dy*ny)
    // -- 4 transcendentals (2 exp, 1 rcp, 1 log).  It is the same real function; in fp32 it differs from EXACT where
    // the reference's subtraction 1-s loses bits, i.e. by ~6e-8*e^y relative on (1-s): a wave takes FAST only when
    // every column of its tile has -20 <= p, -20 <= n <= 3 (a, b are in (0,1), so |x| <= |p|, |y| <= |n|): no
    // overflow of the products, and EXACT/FAST agree to ~1e-6 relative per term, below the 1e-5 loss tolerance.
    auto run_tile = [&](auto fast_t) {
        constexpr bool FAST = decltype(fast_t)::value;
#pragma unroll 2
        for (int k = 0; k < 64; ++k) {
            bool colok = true;
            if (!FULL) colok = cb * 256 + wid * 64 + ((lane + k) & 63) < B;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const v2f z = cn * abs_[q];                                    // -log2(e) * {x, y}
                v2f lg, g;
                if (FAST) {
                    const v2f e2 = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};   // e^-x, e^-y
                    const v2f dd = e2 + one;
                    const v2f nn = __builtin_elementwise_fma(eps, dd, v2f{1.0f, e2.y});
                    const v2f A = dd * nn;
                    const float r = __builtin_amdgcn_rcpf(A.x * A.y);
                    const float N = nn.x * nn.y;
                    lg = v2f{__builtin_amdgcn_logf((N * N) * r), 0.f};
                    const v2f er = e2 * r;
                    g = v2f{er.x * A.y, er.y * A.x};
                } else {
                    const v2f dd = v2f{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + one;
                    const v2f s = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};   // sigmoid(x), sigmoid(y)
                    const v2f om = one - s;
#ifdef MACR_ABL_BXB_8TRANS
                    const v2f tt = v2f{s.x, om.y} + eps;                      // sig(x)+eps, (1-sig(y))+eps
                    lg = v2f{__builtin_amdgcn_logf(tt.x), __builtin_amdgcn_logf(tt.y)};
                    const v2f rt = {__builtin_amdgcn_rcpf(tt.x), __builtin_amdgcn_rcpf(tt.y)};
                    g = (s * om) * rt;
#else
                    // tx = sig(x)+eps, ty = (1-sig(y))+eps.  Only log(tx)+log(ty) and the two reciprocals are needed:
                    // ONE logarithm and ONE reciprocal of the product tx*ty (>= 1e-20, no underflow) serve both
                    // halves: 1/tx = ty/(tx*ty).
                    const v2f ts = s + eps, tom = om + eps;                   // .x of ts and .y of tom are used
                    const float txy = ts.x * tom.y;
                    lg = v2f{__builtin_amdgcn_logf(txy), 0.f};
                    const float r2 = __builtin_amdgcn_rcpf(txy);
                    const v2f h = (s * om) * r2;
                    g = v2f{h.x * tom.y, h.y * ts.x};
#endif
                }
                // g = {-f'(x), g'(y)}: f'(x) = -s(1-s)/(s+eps), g'(y) = s(1-s)/((1-s)+eps); the sign of the x half
                // is applied once, after the loops
                if (!FULL) {
                    const bool ok = rok[q] && colok;
                    lg = ok ? lg : v2f{0.f, 0.f};
                    g = ok ? g : v2f{0.f, 0.f};
                }
                l2 += lg;
                dab[q] = __builtin_elementwise_fma(g, cn, dab[q]);
                acc = __builtin_elementwise_fma(g, ab[q], acc);
            }
            cn.x = wave_rol1(cn.x); cn.y = wave_rol1(cn.y); acc.x = wave_rol1(acc.x); acc.y = wave_rol1(acc.y);
        }
    };
    // FAST with the rows of a lane taken in PAIRS (R = 2, 4): a packed register holds the SAME quantity of two rows instead
    // of the x and y halves of one pair, so every instruction of the chain is packed with both halves useful -- 17 packed,
    // 2 plain, 7 transcendental per TWO pairs (PMC, profiles/pmc_sq_latest.json: a plain or packed VALU instruction is 4
    // issue cycles, a transcendental 8, and the kernel's VALU is busy for all of its time: instruction count is the bound).
    // In this form's range (y <= 6) e^-y + eps*(1 + e^-y) is e^-y to 4e-8 relative (eps/e^-y), so ny = ey and
    //     r = 1/(dx nx dy),  -f'(x) = ex dy r,  g'(y) = dx nx r (= 1/dy),  tx ty = nx^2 ey r,
    // and one logarithm serves the product of the two rows' tx ty (each >= 5e-12: tx >= sig(-20), ty >= 1 - sig(6)).
    auto run_tile_pairs = [&]() {
        constexpr int H = R / 2 > 0 ? R / 2 : 1;
        v2f sa[H], sb[H], a2[H], b2[H], dax[H], dby[H];
        bool ok0[H], ok1[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            a2[h] = v2f{ab[2 * h].x, ab[(2 * h + 1) % R].x}; b2[h] = v2f{ab[2 * h].y, ab[(2 * h + 1) % R].y};
            sa[h] = a2[h] * (-kLog2e); sb[h] = b2[h] * (-kLog2e);
            dax[h] = v2f{0.f, 0.f}; dby[h] = v2f{0.f, 0.f};
            ok0[h] = rok[2 * h]; ok1[h] = rok[(2 * h + 1) % R];
        }
        v2f accx = {0.f, 0.f}, accy = {0.f, 0.f};
        float lsum = 0.f;
#pragma unroll 2
        for (int k = 0; k < 64; ++k) {
            bool colok = true;
            if (!FULL) colok = cb * 256 + wid * 64 + ((lane + k) & 63) < B;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const v2f zx = sa[h] * cn.x, zy = sb[h] * cn.y;
                const v2f ex = {__builtin_amdgcn_exp2f(zx.x), __builtin_amdgcn_exp2f(zx.y)};
                const v2f ey = {__builtin_amdgcn_exp2f(zy.x), __builtin_amdgcn_exp2f(zy.y)};
                const v2f dx = ex + one, dy = ey + one;
                const v2f nx = __builtin_elementwise_fma(eps, dx, one);
                const v2f u = dx * nx;
                const v2f w = u * dy;
                const v2f r = {__builtin_amdgcn_rcpf(w.x), __builtin_amdgcn_rcpf(w.y)};
                const v2f exr = ex * r, eyr = ey * r;
                v2f gx = exr * dy, gy = u * r;
                v2f T = (nx * nx) * eyr;
                if (!FULL) {
                    const bool k0 = ok0[h] && colok, k1 = ok1[h] && colok;
                    gx = v2f{k0 ? gx.x : 0.f, k1 ? gx.y : 0.f}; gy = v2f{k0 ? gy.x : 0.f, k1 ? gy.y : 0.f};
                    T = v2f{k0 ? T.x : 1.0f, k1 ? T.y : 1.0f};
                }
                lsum += __builtin_amdgcn_logf(T.x * T.y);
                dax[h] = __builtin_elementwise_fma(gx, v2f{cn.x, cn.x}, dax[h]);
                dby[h] = __builtin_elementwise_fma(gy, v2f{cn.y, cn.y}, dby[h]);
                accx = __builtin_elementwise_fma(gx, a2[h], accx);
                accy = __builtin_elementwise_fma(gy, b2[h], accy);
            }
            cn.x = wave_rol1(cn.x); cn.y = wave_rol1(cn.y);
            accx.x = wave_rol1(accx.x); accx.y = wave_rol1(accx.y); accy.x = wave_rol1(accy.x); accy.y = wave_rol1(accy.y);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            dab[2 * h] = v2f{dax[h].x, dby[h].x};
            if (2 * h + 1 < R) dab[2 * h + 1] = v2f{dax[h].y, dby[h].y};
        }
        acc = v2f{accx.x + accx.y, accy.x + accy.y};
        l2 = v2f{lsum, 0.f};
    };
#if defined(MACR_ABL_BXB_EXACT)
    const bool fast = false;
#elif defined(MACR_ABL_BXB_FORCEFAST)
    const bool fast = true;
#else
    // The window of the 4-transcendental form is a statement about x = p*a and y = n*b: -20 <= x, YLO <= y <= 3.  a and b
    // are products of sigmoids, in (0,1) and usually well below 1, so the wave bounds them by the maxima over ITS rows:
    // x >= min(p,0)*amax, y <= max(n,0)*bmax, y >= min(n,0)*bmax.  (Columns alone -- a, b bounded by 1 -- put 40-60 % of
    // the tiles of a model a few hundred steps old on the exact side for the 1 % of its negatives that score above 3:
    // profiles/r05_logit_probe.txt.)  YLO: the row-pair form has no e^-2y product and holds down to y = -60.
    float amax = 0.f, bmax = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) { amax = fmaxf(amax, ab[q].x); bmax = fmaxf(bmax, ab[q].y); }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { amax = fmaxf(amax, __shfl_xor(amax, m, 64)); bmax = fmaxf(bmax, __shfl_xor(bmax, m, 64)); }
    // YHI: above it the reference's own rounding is what the exact form reproduces -- fl(1 - fl(sig(y))) is off by up to
    // 3e-8 (1 + e^y) relative, which is this form's distance from it: <= 1.2e-5 of a term that is >= 6 at y = 6, i.e.
    // <= 2e-6 relative on a loss that is a sum of positive terms (tolerance 1e-5); 2e-7 at y = 3, 1.1e-5 at y = 8.
    constexpr float YLO = R >= 2 ? kBxbYLoPairs : kBxbYLoSingle, YHI = kBxbYHi;
    const float xlo = cn.x * amax, yv = cn.y * bmax;
    const bool fast = !__any(!(xlo >= -20.0f) || !(yv >= YLO) || !(yv <= YHI) || !(amax <= 1.0f) || !(bmax <= 1.0f));   // (NaN -> exact path)
#endif
    if (fast) {
        if constexpr (R >= 2) run_tile_pairs(); else run_tile(std::true_type{});
    } else {
        run_tile(std::false_type{});
    }
    if (FULL && sort.nneu) l2.x += 2.0f * (float)(R * (int)__popcll(__ballot(own_out)));     // the neutral evaluations of this lane: -2 each
    if (cok && !own_out) {                      // home again: column c over this wave's 64*R rows
        colpart[((size_t)rb * 2 + 0) * Bp + c] = -acc.x;
        colpart[((size_t)rb * 2 + 1) * Bp + c] = acc.y;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        s_row[wid][0][q * 64 + lane] = -dab[q].x;
        s_row[wid][1][q * 64 + lane] = dab[q].y;
    }
    const float lsum = block_sum(-(l2.x + l2.y) * kLn2, red);   // contains the barrier that publishes s_row
    if (t == 0) lpart[(size_t)rb * ncbx + cb] = lsum;
    for (int e = t; e < 2 * RB; e += 256) {
        const int q = e / RB, rr = e % RB, r = rb * RB + rr;
        if (FULL || r < B)
            rowpart[((size_t)cb * 2 + q) * Bp + r] = (s_row[0][q][rr] + s_row[1][q][rr]) + (s_row[2][q][rr] + s_row[3][q][rr]);
    }
}

// ----------------------------------------------------------------------------
// Row scatter pattern (measured, tools/atomic_bench.hip): a device-scope fp32 atomic INSTRUCTION
// that hits an L2 line already being updated serialises at ~24 ns.  One wave therefore owns one
// whole row: lane k adds element k (+64, +128 ...), so a 256-B row at d=64 is ONE atomic
// instruction covering two full lines instead of four instructions touching both lines four
// times (16 lanes x float4 layout): 45 -> 12.5 us for a Zipf batch of 12288 row references.
// Branch-vector gradients are NOT accumulated with atomics at all (every block would hit the same
// d addresses); each block writes one partial row that adam_dense reduces.
// ----------------------------------------------------------------------------
constexpr int kChunkT = 16;       // consecutive triples per block pass in the backward kernels

template <int D>
struct WaveRow {
    static constexpr int EPL = (D + 63) / 64;          // elements per lane
    static constexpr int kActive = D < 64 ? D : 64;    // active lanes (d=32 leaves half the wave idle)
};

// The positive rows of a chunk (s_pos[slot] = row, -1 past the end of the batch; s_gi[slot] = its gradient row):
// thread k owns element k and adds every DISTINCT row once -- equal rows of the chunk are summed first, adjacent or
// not (the batch arrives bucketed by the low byte of the positive item, batch_bucket_block, so the hottest item of a
// batch costs (#chunks it spans) serialised atomics instead of (#references)).
template <int D>
__device__ __forceinline__ void combine_positive_rows(const int *s_pos, const float (*s_gi)[D], float *gI,
                                                      int32_t *cnt = nullptr) {
    // cnt (LightGCN): cnt[row] += the row's references in this chunk, one atomic per DISTINCT row of the chunk
    // thread = (element k, part): part p of NP handles the distinct rows whose FIRST slot L has L % NP == p
    constexpr int NP = D >= 256 ? 1 : 256 / D;
    const int k = threadIdx.x % D, part = threadIdx.x / D;
    int rows[kChunkT];
#pragma unroll
    for (int s = 0; s < kChunkT; ++s) rows[s] = s_pos[s];
#pragma unroll
    for (int L = 0; L < kChunkT; ++L) {
        if (rows[L] < 0 || (L % NP) != part) continue;
        bool first = true;
#pragma unroll
        for (int s0 = 0; s0 < L; ++s0) first = first && rows[s0] != rows[L];
        if (!first) continue;
        float acc = s_gi[L][k];
        int mult = 1;
#pragma unroll
        for (int s2 = L + 1; s2 < kChunkT; ++s2)
            if (rows[s2] == rows[L]) { acc += s_gi[s2][k]; ++mult; }
        MACR_ATOMIC_ADD(gI + (size_t)rows[L] * D + k, acc);
        if (cnt && k == 0) atomicAdd(cnt + rows[L], mult);
    }
}

// ----------------------------------------------------------------------------
// pair_bwd (rubibceboth): gradient rows + scatter-add.           SURVEY.md A.1
//   dp,dn (column sums) and da,db (row sums) come from the bxb partials / B^2
//   dsi=da*sig'(si)*sig(su)+(alpha/B)f'(si) ...  deu=dp*ei+dn*ej+dsu*wu+coef*eu ...
//   gU[u]+=deu, gI[i]+=dei, gI[j]+=dej  (duplicates summed = TF IndexedSlices
//   de-duplication before the sparse apply, macr_mf/model.py:74)
//   wpart[block % kBranchSlots] += sum over the block's triples of {ei*dsi+ej*dsj, eu*dsu}
//   (a few partial rows instead of one: 32 blocks per slot add without queueing on one line)
// One wave per triple (grid-strided); lane k owns element k of every row.
// ----------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_pair_bwd(
    int B, int Bp, int nrb, int ncb, const int32_t *__restrict__ perm, const int32_t *__restrict__ u,
    const int32_t *__restrict__ i, const int32_t *__restrict__ j, const float *__restrict__ Usrc,
    const float *__restrict__ Isrc,
    const float *__restrict__ w, const float *__restrict__ wu, const float *__restrict__ fwd,
    const float *__restrict__ rowpart, const float *__restrict__ colpart,
    float *gU, float *gI, int32_t *touchedU, int32_t *touchedI, float *__restrict__ wpart,
    float alpha, float beta, float coef, float *adam_pow, StepScalars *scal, float lr, float b1, float b2,
    LossArgs L, int32_t *cnt_pos, LazyState *lazy = nullptr, int nneu = 0) {
    constexpr int EPL = WaveRow<D>::EPL;
    __shared__ float s_w[4][2][D];
    __shared__ float s_gi[kChunkT][D];
    __shared__ int s_pos[kChunkT];
    const int nblk = gridDim.x - 1;
    if ((int)blockIdx.x == nblk) {
        // Step bookkeeping, by one wave of an extra block.  Every Adam pass of the PREVIOUS update has completed
        // before this kernel starts and every pass of THIS update starts after it ends, so this is where lr_t may
        // change: Adam bias correction for this step, then advance TF's fp32 beta powers.
        if (threadIdx.x == 0) {
            const float p1 = adam_pow[0], p2 = adam_pow[1];
            const float lr_t = lr * sqrtf(1.0f - p2) / (1.0f - p1);
            scal->lr_t = lr_t;
            adam_pow[0] = p1 * b1;
            adam_pow[1] = p2 * b2;
            if (lazy) lazy_tick(lazy, lr_t);
        }
        if (L.losses && threadIdx.x < 64) finalize_losses(L, threadIdx.x);   // deferred mode: no adam_dense launch to do it
        return;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool act = lane < WaveRow<D>::kActive;
    float wk[EPL], wuk[EPL], aw[EPL], awu[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int k = lane + 64 * e;
        wk[e] = act ? w[k] : 0.f; wuk[e] = act ? wu[k] : 0.f; aw[e] = 0.f; awu[e] = 0.f;
    }
    const float inv_b2 = 1.0f / ((float)B * (float)B), eps = 1e-10f, invB = 1.0f / (float)B;
    constexpr int SPW = kChunkT / 4;               // slots per wave
    // A block owns kChunkT CONSECUTIVE slots of the batch BUCKETED BY POSITIVE ITEM (u, i, j are the bucketed copies,
    // perm[s] the triple's position in the caller's batch = its index in fwd and in the bxb partials; see
    // batch_bucket_block) and adds equal positive rows of a chunk once (combine_positive_rows).  A wave owns SPW of
    // the slots from the index loads to the atomics: one trip for the indices, one for everything that depends on
    // them (bxb partials, forward scalars, rows), no barrier in between.
    for (int chunk = blockIdx.x; chunk * kChunkT < B; chunk += nblk) {
        const int slot0 = chunk * kChunkT + wid * SPW;
        int my_idx = 0;                             // lanes 0..4*SPW-1: u | i | j | perm of the wave's slots
        {
            const int tq = slot0 + (lane % SPW);
            if (lane < 4 * SPW && tq < B) my_idx = (lane < SPW ? u : lane < 2 * SPW ? i : lane < 3 * SPW ? j : perm)[tq];
        }
        // Everything that depends on the indices goes out in ONE straight-line burst (clamped addresses, no branch between the
        // loads): the wave's SPW x 3 rows first, then the bxb partials and the forward scalars.  (Round 6: the ISA of the
        // version before showed what its source did not -- a wait inside each partial loop, then per slot "load three rows,
        // wait, atomics": eight dependent trips per wave instead of two; exec-mask branches end basic blocks and the
        // compiler hoists no load over them.)
        float reu[SPW][EPL], rei[SPW][EPL], rej[SPW][EPL];
        int rru[SPW], rri[SPW], rrj[SPW];
#pragma unroll
        for (int q = 0; q < SPW; ++q) {             // (slots behind the end of the batch hold index 0: row 0, loaded and ignored)
            rru[q] = __shfl(my_idx, q, kWave); rri[q] = __shfl(my_idx, SPW + q, kWave); rrj[q] = __shfl(my_idx, 2 * SPW + q, kWave);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int k = act ? lane + 64 * e : 0;
                reu[q][e] = Usrc[(size_t)rru[q] * D + k]; rei[q][e] = Isrc[(size_t)rri[q] * D + k]; rej[q][e] = Isrc[(size_t)rrj[q] * D + k];
            }
        }
        // column sums dp,dn and row sums da,db of the (B,B) term: lane (q = lane/16, kk = lane%16) adds partials
        // kk, kk+16, ... of slot q, then the 16 lanes of a slot meet by shuffles
        const int q_own = lane >> 4, kk = lane & 15;
        float dp = 0.f, dn = 0.f, da = 0.f, db = 0.f, ssi = 0.f, ssj = 0.f, ssu = 0.f;
        const bool own_ok = slot0 + q_own < B;
        const int t_own = own_ok ? __shfl(my_idx, 3 * SPW + q_own, kWave) : 0;
        {
            // up to 64 row blocks and 32 column blocks unrolled with clamped indices (every small-batch shape: B <= 8192 at
            // R = 4 is nrb <= 32, R = 1 below 4096 is nrb <= 64); a loop behind them for forced shapes beyond
            float cp[4], cn[4], ra[2], rb[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = kk + 16 * m, kc = k < nrb ? k : 0;
                cp[m] = colpart[((size_t)kc * 2) * Bp + t_own]; cn[m] = colpart[((size_t)kc * 2 + 1) * Bp + t_own];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int k = kk + 16 * m, kc = k < ncb ? k : 0;
                ra[m] = rowpart[((size_t)kc * 2) * Bp + t_own]; rb[m] = rowpart[((size_t)kc * 2 + 1) * Bp + t_own];
            }
            // (the neutral blocks' slab of row sums -- k_bxb -- is slab number ncb .. ncb + nneu - 1)
            const int kn = kk < nneu ? ncb + kk : 0;
            const float dan = rowpart[((size_t)kn * 2) * Bp + t_own], dbn = rowpart[((size_t)kn * 2 + 1) * Bp + t_own];
            const float sg = fabsf(fwd[(4 + (kk < 3 ? kk : 0)) * (size_t)Bp + t_own]);   // lane kk = 0,1,2: sig(si), sig(sj), |sig(su)| (its sign: a flag of pair_fwd)
#pragma unroll
            for (int m = 0; m < 4; ++m) { const bool ok = own_ok && kk + 16 * m < nrb; dp += ok ? cp[m] : 0.f; dn += ok ? cn[m] : 0.f; }
#pragma unroll
            for (int m = 0; m < 2; ++m) { const bool ok = own_ok && kk + 16 * m < ncb; da += ok ? ra[m] : 0.f; db += ok ? rb[m] : 0.f; }
            const bool neu_ok = own_ok && kk < nneu;               // (selects, not branches: a load whose only use sits under a branch is
            da += neu_ok ? dan : 0.f; db += neu_ok ? dbn : 0.f;    // sunk into it by the compiler and becomes a trip of its own)
            ssi = (own_ok && kk < 3) ? sg : 0.f;
            if (own_ok) {                           // (wave-uniform trip counts: shapes beyond the unrolled part)
                for (int k = kk + 64; k < nrb; k += 16) { dp += colpart[((size_t)k * 2) * Bp + t_own]; dn += colpart[((size_t)k * 2 + 1) * Bp + t_own]; }
                for (int k = kk + 32; k < ncb; k += 16) { da += rowpart[((size_t)k * 2) * Bp + t_own]; db += rowpart[((size_t)k * 2 + 1) * Bp + t_own]; }
            }
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            dp += __shfl_xor(dp, m, kWave); dn += __shfl_xor(dn, m, kWave);
            da += __shfl_xor(da, m, kWave); db += __shfl_xor(db, m, kWave);
        }
        ssj = __shfl(ssi, (lane & 48) + 1, kWave); ssu = __shfl(ssi, (lane & 48) + 2, kWave); ssi = __shfl(ssi, lane & 48, kWave);
        dp *= inv_b2; dn *= inv_b2; da *= inv_b2; db *= inv_b2;
        const float dsi_own = da * (ssi * (1.0f - ssi)) * ssu + (alpha * invB) * dneglog_sig(ssi, eps);
        const float dsj_own = db * (ssj * (1.0f - ssj)) * ssu + (alpha * invB) * dneglog_1msig(ssj, eps);
        const float dsu_own = (da * ssi + db * ssj) * (ssu * (1.0f - ssu)) +
                              (beta * invB) * (dneglog_sig(ssu, eps) + dneglog_1msig(ssu, eps));
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const int slot = wid * SPW + q;
            if (slot0 + q >= B) { if (lane == 0) s_pos[slot] = -1; continue; }
            const float dpq = __shfl(dp, q * 16, kWave), dnq = __shfl(dn, q * 16, kWave);
            const float dsi = __shfl(dsi_own, q * 16, kWave), dsj = __shfl(dsj_own, q * 16, kWave), dsu = __shfl(dsu_own, q * 16, kWave);
            const int ru = rru[q], ri = rri[q], rj = rrj[q];
            if (act) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int k = lane + 64 * e;
                    const float eu = reu[q][e], ei = rei[q][e], ej = rej[q][e];
                    const float gu = fmaf(coef, eu, fmaf(dsu, wuk[e], fmaf(dnq, ej, dpq * ei)));
                    const float gi = fmaf(coef, ei, fmaf(dsi, wk[e], dpq * eu));
                    const float gj = fmaf(coef, ej, fmaf(dsj, wk[e], dnq * eu));
                    MACR_ATOMIC_ADD(gU + (size_t)ru * D + k, gu);
                    MACR_ATOMIC_ADD(gI + (size_t)rj * D + k, gj);
                    s_gi[slot][k] = gi;                         // positive row: combined per chunk below
                    aw[e] = fmaf(dsi, ei, fmaf(dsj, ej, aw[e]));
                    awu[e] = fmaf(dsu, eu, awu[e]);
                }
            }
            if (lane == 0) {
                s_pos[slot] = ri;
                if (touchedU) { touchedU[ru] = 1; touchedI[ri] = 1; touchedI[rj] = 1; }
            }
        }
        __syncthreads();
        combine_positive_rows<D>(s_pos, s_gi, gI, cnt_pos);
        __syncthreads();
    }
    if (act) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) { s_w[wid][0][lane + 64 * e] = aw[e]; s_w[wid][1][lane + 64 * e] = awu[e]; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * D; k += 256) {
        const int q = k / D, kk2 = k % D;
        MACR_ATOMIC_ADD(wpart + (size_t)(blockIdx.x % kBranchSlots) * 2 * D + k,
                        (s_w[0][q][kk2] + s_w[1][q][kk2]) + (s_w[2][q][kk2] + s_w[3][q][kk2]));
    }
}

// ----------------------------------------------------------------------------
// pair_normal: `normalbce` forward + backward in one pass (genuinely per pair).
//   mf = mean(-log(sig(p)+1e-9) - log(1-sig(n)+1e-9))            macr_mf/model.py:277-287
//   dp = f'(p)/B, dn = g'(n)/B;  deu=dp*ei+dn*ej, dei=dp*eu, dej=dn*eu (+coef*row)
// Algorithmic HBM bytes per triple: 3 rows read + 3 gradient rows written + 12 B indices
// = 24*d+12.  One wave per triple, lane k owns element k.
// ----------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_pair_normal(
    int B, const int32_t *__restrict__ u, const int32_t *__restrict__ i, const int32_t *__restrict__ j,
    const float *__restrict__ Usrc, const float *__restrict__ Isrc,
    float *gU, float *gI, int32_t *touchedU, int32_t *touchedI, float *__restrict__ part, float coef,
    int reg_on_gathered, const float *__restrict__ adam_pow_in, float *adam_pow_out, StepScalars *scal,
    float lr, float b1, float b2, int32_t *cnt_pos) {
    constexpr int EPL = WaveRow<D>::EPL;
    __shared__ float red[16];
    __shared__ float s_gi[kChunkT][D];
    __shared__ int s_pos[kChunkT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool act = lane < WaveRow<D>::kActive;
    if (blockIdx.x == 0 && threadIdx.x == 0 && adam_pow_out) {      // (NULL: loss-only pass, no step is taken)
        const float p1 = adam_pow_in[0], p2 = adam_pow_in[1];
        scal->lr_t = lr * sqrtf(1.0f - p2) / (1.0f - p1);
        adam_pow_out[0] = p1 * b1;
        adam_pow_out[1] = p2 * b2;
    }
    float sq = 0.f, bce = 0.f;
    const float eps = 1e-9f, invB = 1.0f / (float)B;
    for (int chunk = blockIdx.x; chunk * kChunkT < B; chunk += gridDim.x) {      // see pair_bwd: equal positives of a chunk are added once
#pragma unroll 1
        for (int q = 0; q < kChunkT / 4; ++q) {
            const int slot = wid * (kChunkT / 4) + q;
            const int t = chunk * kChunkT + slot;
            if (t >= B) { if (lane == 0) s_pos[slot] = -1; continue; }
            const int ru = u[t], ri = i[t], rj = j[t];
            float eu[EPL], ei[EPL], ej[EPL], pp = 0.f, nn = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int k = lane + 64 * e;
                eu[e] = act ? Usrc[(size_t)ru * D + k] : 0.f;
                ei[e] = act ? Isrc[(size_t)ri * D + k] : 0.f;
                ej[e] = act ? Isrc[(size_t)rj * D + k] : 0.f;
                pp = fmaf(eu[e], ei[e], pp); nn = fmaf(eu[e], ej[e], nn);
                if (reg_on_gathered) sq += eu[e] * eu[e] + ei[e] * ei[e] + ej[e] * ej[e];
            }
            const float p = wave_sum(pp), n = wave_sum(nn);
            const float sp = sigmoid_acc(p), sn = sigmoid_acc(n);
            if (lane == 0) bce += -logf(sp + eps) + -logf((1.0f - sn) + eps);
            const float dp = dneglog_sig(sp, eps) * invB, dn = dneglog_1msig(sn, eps) * invB;
            if (act && gU) {                                    // gU == NULL: forward only (loss-only pass)
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int k = lane + 64 * e;
                    MACR_ATOMIC_ADD(gU + (size_t)ru * D + k, fmaf(coef, eu[e], fmaf(dn, ej[e], dp * ei[e])));
                    MACR_ATOMIC_ADD(gI + (size_t)rj * D + k, fmaf(coef, ej[e], dn * eu[e]));
                    s_gi[slot][k] = fmaf(coef, ei[e], dp * eu[e]);
                }
            }
            if (lane == 0) {
                s_pos[slot] = ri;
                if (touchedU) { touchedU[ru] = 1; touchedI[ri] = 1; touchedI[rj] = 1; }
            }
        }
        __syncthreads();
        if (gI) combine_positive_rows<D>(s_pos, s_gi, gI, cnt_pos);
        __syncthreads();
    }
    const float s0 = block_sum(sq, red);
    const float s3 = block_sum(bce, red);
    if (threadIdx.x == 0) {
        float *o = part + (size_t)blockIdx.x * kPartStride;
        o[0] = s0; o[1] = 0.f; o[2] = 0.f; o[3] = s3;
    }
}

// ============================================================================
// Large batches: staging + sorted references + segment reduce (no atomics on the path).
// Above kSmallBatchMax triples the pair kernels stop being latency bound and become HBM bound, and a row atomic is a
// read-modify-write at L2/HBM (measured: 1.44x the algorithmic traffic, 0.85 TB/s at B = 2^20).  Here the pair
// kernel reads its 3 rows once and writes its 3 gradient rows ONCE, with plain coalesced stores, to a staging buffer
// in batch order -- exactly the algorithmic 24*d+12 bytes per triple -- while the 3B references (key = table row,
// value = staging row) are radix-sorted by row (refsort.hpp).  k_seg_reduce then gives every row ONE owner that
// sums its contributions in sorted (= batch) order and stores the row into gP/gQ: the IndexedSlices de-duplication
// of tf.train.AdamOptimizer (macr_mf/model.py:74), deterministic.  Rows referenced more than kSegChunk times are cut
// into chunks that add atomically (a handful of atomics for the hottest rows only).
// ============================================================================
constexpr int kSmallBatchMax = 8192;

__global__ __launch_bounds__(256) void k_refs_init(int B, int n_users, const int32_t *__restrict__ u,
                                                   const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                   uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                   uint32_t *__restrict__ n_work) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_work = 0u;           // work list of k_seg_scan, empty again
    for (int t = blockIdx.x * 256 + threadIdx.x; t < B; t += gridDim.x * 256) {
        key[t] = (uint32_t)u[t];                       val[t] = (uint32_t)t;
        key[B + t] = (uint32_t)(n_users + i[t]);       val[B + t] = (uint32_t)(B + t);
        key[2 * (size_t)B + t] = (uint32_t)(n_users + j[t]);   val[2 * (size_t)B + t] = (uint32_t)(2 * (size_t)B + t);
    }
}

// normalbce forward+backward, one LPR-lane group per triple, float4 per lane; see k_pair_normal for the arithmetic.
// stage: [3][B][d] gradient rows of (user, positive, negative) in batch order.
template <int LPR>
__global__ __launch_bounds__(256) void k_pair_normal_stage(
    int B, const int32_t *__restrict__ u, const int32_t *__restrict__ i, const int32_t *__restrict__ j,
    const float *__restrict__ Usrc, const float *__restrict__ Isrc, float *__restrict__ stage,
    float *__restrict__ part, float coef, int reg_on_gathered, const float *__restrict__ adam_pow_in,
    float *adam_pow_out, StepScalars *scal, float lr, float b1, float b2, const uint32_t *__restrict__ place = nullptr) {
    constexpr int d = 4 * LPR, RPB = RowGroup<LPR>::kRowsPerBlock;
    __shared__ float red[16];
    RowGroup<LPR> g;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float p1 = adam_pow_in[0], p2 = adam_pow_in[1];
        scal->lr_t = lr * sqrtf(1.0f - p2) / (1.0f - p1);
        adam_pow_out[0] = p1 * b1;
        adam_pow_out[1] = p2 * b2;
    }
    float sq = 0.f, bce = 0.f;
    const float eps = 1e-9f, invB = 1.0f / (float)B;
    for (long long base = (long long)blockIdx.x * RPB; base < B; base += (long long)gridDim.x * RPB) {
        const long long t = base + g.slot;
        if (t >= B) continue;
        const int ru = u[t], ri = i[t], rj = j[t];
        const float4 eu = ld4(Usrc + (size_t)ru * d + 4 * g.sub);
        const float4 ei = ld4(Isrc + (size_t)ri * d + 4 * g.sub);
        const float4 ej = ld4(Isrc + (size_t)rj * d + 4 * g.sub);
        const float p = group_sum<LPR>(dot4(eu, ei)), n = group_sum<LPR>(dot4(eu, ej));
        if (reg_on_gathered) sq += dot4(eu, eu) + dot4(ei, ei) + dot4(ej, ej);
        const float sp = sigmoid_acc(p), sn = sigmoid_acc(n);
        if (g.sub == 0) bce += -logf(sp + eps) + -logf((1.0f - sn) + eps);
        const float dp = dneglog_sig(sp, eps) * invB, dn = dneglog_1msig(sn, eps) * invB;
        // place (may be NULL): where reference (role, t) stands in the list sorted by row -- its gradient row goes THERE,
        // so that a row's references are consecutive staging rows for whoever sums them
        const size_t o0 = place ? place[t] : (size_t)t, o1 = place ? place[(size_t)B + t] : (size_t)B + t;
        const size_t o2 = place ? place[2 * (size_t)B + t] : 2 * (size_t)B + t;
        st4(stage + o0 * d + 4 * g.sub, fma4(coef, eu, fma4(dn, ej, scale4(dp, ei))));
        st4(stage + o1 * d + 4 * g.sub, fma4(coef, ei, scale4(dp, eu)));
        st4(stage + o2 * d + 4 * g.sub, fma4(coef, ej, scale4(dn, eu)));
    }
    const float s0 = block_sum(sq, red);
    const float s3 = block_sum(bce, red);
    if (threadIdx.x == 0) {
        float *o = part + (size_t)blockIdx.x * kPartStride;
        o[0] = s0; o[1] = 0.f; o[2] = 0.f; o[3] = s3;
    }
}

// rubibceboth backward; see k_pair_bwd for the arithmetic.  The last block does the step bookkeeping.
template <int LPR>
__global__ __launch_bounds__(256) void k_pair_bwd_stage(
    int B, int Bp, int nrb, int ncb, const int32_t *__restrict__ u, const int32_t *__restrict__ i,
    const int32_t *__restrict__ j, const float *__restrict__ Usrc, const float *__restrict__ Isrc,
    const float *__restrict__ w, const float *__restrict__ wu, const float *__restrict__ fwd,
    const float *__restrict__ rowpart, const float *__restrict__ colpart, float *__restrict__ stage,
    float *__restrict__ wpart, float alpha, float beta, float coef, float *adam_pow, StepScalars *scal, float lr,
    float b1, float b2, LossArgs L, const uint32_t *__restrict__ place = nullptr, int Bnorm = 0, LazyState *lazy = nullptr,
    int nneu = 0) {
    // Bnorm > 0: the launch covers a SLICE of a batch of Bnorm triples (row-sharded training, macr_shard_backward_slice): u/i/j,
    // fwd, rowpart and colpart arrive offset to the slice, B is its length, the means are taken over the whole batch
    constexpr int d = 4 * LPR, RPB = RowGroup<LPR>::kRowsPerBlock;
    __shared__ float4 s_w[2][256];
    const int nblk = gridDim.x - 1;
    if ((int)blockIdx.x == nblk) {
        if (threadIdx.x == 0) {
            const float p1 = adam_pow[0], p2 = adam_pow[1];
            const float lr_t = lr * sqrtf(1.0f - p2) / (1.0f - p1);
            scal->lr_t = lr_t;
            adam_pow[0] = p1 * b1;
            adam_pow[1] = p2 * b2;
            if (lazy) lazy_tick(lazy, lr_t);
        }
        if (L.losses && threadIdx.x < 64) finalize_losses(L, threadIdx.x);
        return;
    }
    RowGroup<LPR> g;
    const float4 w4 = ld4(w + 4 * g.sub), wu4 = ld4(wu + 4 * g.sub);
    float4 aw = make_float4(0, 0, 0, 0), awu = make_float4(0, 0, 0, 0);
    const float Bn = (float)(Bnorm > 0 ? Bnorm : B);
    const float inv_b2 = 1.0f / (Bn * Bn), eps = 1e-10f, invB = 1.0f / Bn;
    for (long long base = (long long)blockIdx.x * RPB; base < B; base += (long long)nblk * RPB) {
        const long long t = base + g.slot;
        if (t >= B) continue;
        float dp = 0.f, dn = 0.f, da = 0.f, db = 0.f;
        for (int k = g.sub; k < nrb; k += LPR) { dp += colpart[((size_t)k * 2) * Bp + t]; dn += colpart[((size_t)k * 2 + 1) * Bp + t]; }
        float dan = 0.f, dbn = 0.f;                 // (the neutral blocks' slab: see k_pair_bwd)
        if (g.sub < nneu) { dan = rowpart[((size_t)(ncb + g.sub) * 2) * Bp + t]; dbn = rowpart[((size_t)(ncb + g.sub) * 2 + 1) * Bp + t]; }
        for (int k = g.sub; k < ncb; k += LPR) { da += rowpart[((size_t)k * 2) * Bp + t]; db += rowpart[((size_t)k * 2 + 1) * Bp + t]; }
        da += dan; db += dbn;
        dp = group_sum<LPR>(dp) * inv_b2; dn = group_sum<LPR>(dn) * inv_b2;
        da = group_sum<LPR>(da) * inv_b2; db = group_sum<LPR>(db) * inv_b2;
        const float ssi = fwd[4 * (size_t)Bp + t], ssj = fwd[5 * (size_t)Bp + t], ssu = fabsf(fwd[6 * (size_t)Bp + t]);    // (the sign: a flag of pair_fwd)
        const float dsi = da * (ssi * (1.0f - ssi)) * ssu + (alpha * invB) * dneglog_sig(ssi, eps);
        const float dsj = db * (ssj * (1.0f - ssj)) * ssu + (alpha * invB) * dneglog_1msig(ssj, eps);
        const float dsu = (da * ssi + db * ssj) * (ssu * (1.0f - ssu)) +
                          (beta * invB) * (dneglog_sig(ssu, eps) + dneglog_1msig(ssu, eps));
        const int ru = u[t], ri = i[t], rj = j[t];
        const float4 eu = ld4(Usrc + (size_t)ru * d + 4 * g.sub);
        const float4 ei = ld4(Isrc + (size_t)ri * d + 4 * g.sub);
        const float4 ej = ld4(Isrc + (size_t)rj * d + 4 * g.sub);
        const size_t o0 = place ? place[t] : (size_t)t, o1 = place ? place[(size_t)B + t] : (size_t)B + t;
        const size_t o2 = place ? place[2 * (size_t)B + t] : 2 * (size_t)B + t;       // (see k_pair_normal_stage)
        st4(stage + o0 * d + 4 * g.sub, fma4(coef, eu, fma4(dsu, wu4, fma4(dn, ej, scale4(dp, ei)))));
        st4(stage + o1 * d + 4 * g.sub, fma4(coef, ei, fma4(dsi, w4, scale4(dp, eu))));
        st4(stage + o2 * d + 4 * g.sub, fma4(coef, ej, fma4(dsj, w4, scale4(dn, eu))));
        aw = fma4(dsi, ei, fma4(dsj, ej, aw));
        awu = fma4(dsu, eu, awu);
    }
    s_w[0][threadIdx.x] = aw; s_w[1][threadIdx.x] = awu;
    __syncthreads();
    if (threadIdx.x < 2 * LPR) {                       // thread (q, sub): sums its float4 over the block's row groups
        const int q = threadIdx.x / LPR, sub = threadIdx.x % LPR;
        float4 acc = make_float4(0, 0, 0, 0);
        for (int k = 0; k < RPB; ++k) acc = add4(acc, s_w[q][k * LPR + sub]);
        float *dst = wpart + (size_t)(blockIdx.x % kBranchSlots) * 2 * d + (size_t)q * d + 4 * sub;
        MACR_ATOMIC_ADD(dst + 0, acc.x); MACR_ATOMIC_ADD(dst + 1, acc.y);
        MACR_ATOMIC_ADD(dst + 2, acc.z); MACR_ATOMIC_ADD(dst + 3, acc.w);
    }
}

// One LPR-lane group per position p of the sorted reference list.  The group at the head of a chunk -- first
// reference of a row, or a multiple of CH -- sums the (at most CH) staging rows of its chunk in list order and
// stores the row (whole run in one chunk: plain store; gP/gQ rows are zero between steps) or adds it atomically
// (runs cut into several chunks).
template <int LPR>
__global__ __launch_bounds__(256) void k_seg_reduce(int n, int n_users, uint32_t key_end, const uint32_t *__restrict__ sk,
                                                    const uint32_t *__restrict__ sv, const float *__restrict__ stage,
                                                    float *gU, float *gI, int32_t *tU, int32_t *tI) {
    constexpr int d = 4 * LPR, RPB = RowGroup<LPR>::kRowsPerBlock, CH = LPR < 16 ? LPR : 16;
    RowGroup<LPR> g;
    const long long p = (long long)blockIdx.x * RPB + g.slot;
    const bool in = p < n;
    const int lane = threadIdx.x & 63, gbase = lane - g.sub;
    uint32_t ks = 0xffffffffu, vs = 0u;
    if (in && g.sub < CH && p + g.sub < n) { ks = sk[p + g.sub]; vs = sv[p + g.sub]; }
    const uint32_t key0 = __shfl(ks, gbase, kWave);
    const uint32_t prev = (in && p > 0) ? sk[p - 1] : 0xffffffffu;
    const uint64_t eq = __ballot(in && g.sub < CH && ks == key0);
    const uint64_t gmask = LPR == 64 ? ~0ull : (((1ull << LPR) - 1ull) << gbase);
    const int cnt_eq = __popcll(eq & gmask);                        // equal keys are contiguous from the head (sorted)
    const int lim = (int)(p % CH) == 0 ? CH : CH - (int)(p % CH);
    const int len = cnt_eq < lim ? cnt_eq : lim;
    const bool head = in && key0 < key_end && (key0 != prev || (p % CH) == 0);     // keys >= key_end: not this rank's rows
    if (head) {
        const uint32_t next = (p + len < n) ? sk[p + len] : 0xffffffffu;
        const bool multi = key0 == prev || next == key0;
        const bool is_user = key0 < (uint32_t)n_users;
        const size_t row = is_user ? key0 : key0 - (uint32_t)n_users;
        float4 acc = make_float4(0, 0, 0, 0);
        int k = 0;
        for (; k + 4 <= len; k += 4) {
            const uint32_t r0 = __shfl(vs, gbase + k, kWave), r1 = __shfl(vs, gbase + k + 1, kWave);
            const uint32_t r2 = __shfl(vs, gbase + k + 2, kWave), r3 = __shfl(vs, gbase + k + 3, kWave);
            const float4 x0 = ld4(stage + (size_t)r0 * d + 4 * g.sub), x1 = ld4(stage + (size_t)r1 * d + 4 * g.sub);
            const float4 x2 = ld4(stage + (size_t)r2 * d + 4 * g.sub), x3 = ld4(stage + (size_t)r3 * d + 4 * g.sub);
            acc = add4(add4(add4(add4(acc, x0), x1), x2), x3);
        }
        for (; k < len; ++k) {
            const uint32_t r = __shfl(vs, gbase + k, kWave);
            acc = add4(acc, ld4(stage + (size_t)r * d + 4 * g.sub));
        }
        float *dst = (is_user ? gU : gI) + row * d + 4 * g.sub;
        if (multi) {
            MACR_ATOMIC_ADD(dst + 0, acc.x); MACR_ATOMIC_ADD(dst + 1, acc.y);
            MACR_ATOMIC_ADD(dst + 2, acc.z); MACR_ATOMIC_ADD(dst + 3, acc.w);
        } else {
            st4(dst, acc);
        }
        if (g.sub == 0 && key0 != prev && tU) (is_user ? tU : tI)[row] = 1;
    }
}

// The same de-duplication for a step whose Adam pass follows at once (adam_block INDEXED): nothing is summed for a row
// with at most kRefRun references -- its flag names them, (count << kRefShift) | position of the first, and the pass
// sums them.  One THREAD per position looks for the first reference of a row (k_seg_reduce spends a lane group per
// position on that: 238 us at 3 M references even with nothing to sum) and counts the run.  Rows with more references
// (the hot items of a popularity-skewed batch) get flag 1 and their gradient row in gU/gI as before: the thread finds
// the end of the run by galloping + bisection in the sorted keys and appends one work item per kLongSpan references,
// which k_seg_sum adds up, one wave per item.
constexpr int kRefRun = 16;            // <= 31: the count shares the flag word with a position below kRefMaxRefs
constexpr int kLongSpan = 64;

__global__ __launch_bounds__(256) void k_seg_scan(int n, int n_users, uint32_t key_end, const uint32_t *__restrict__ sk,
                                                  int32_t *tU, int32_t *tI, uint32_t *__restrict__ work,
                                                  uint32_t *__restrict__ n_work) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t key = sk[p];
    const uint32_t prev = p > 0 ? sk[p - 1] : 0xffffffffu;
    if (key >= key_end || key == prev) return;                        // not the first reference of one of this rank's rows
    const int lim = kRefRun + 1 < n - p ? kRefRun + 1 : (int)(n - p);
    int len = 1;
    for (bool more = true; more && len < lim;) {                      // four keys per trip: most runs end in the first
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = len + k < lim ? sk[p + len + k] : ~key;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (more && w[k] == key) ++len; else more = false;
        }
    }
    const bool is_user = key < (uint32_t)n_users;
    int32_t *flag = (is_user ? tU : tI) + (is_user ? key : key - (uint32_t)n_users);
    if (len <= kRefRun) {
        *flag = (len << kRefShift) | (int)p;
        return;
    }
    *flag = 1;
    // end of the run: first position q > p with sk[q] != key
    long long lo = p + len, step = kRefRun;                           // sk[lo - 1] == key
    while (lo + step <= n && sk[lo + step - 1] == key) { lo += step; step *= 2; }
    long long hi = lo + step < n ? lo + step : n;                     // sk[hi] != key or hi == n; answer in [lo, hi]
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (sk[mid] == key) lo = mid + 1; else hi = mid;
    }
    const uint32_t total = (uint32_t)(lo - p), items = (total + kLongSpan - 1) / kLongSpan;
    const uint32_t at = atomicAdd(n_work, 2 * items);
    for (uint32_t k = 0; k < items; ++k) {
        const uint32_t span = total - k * kLongSpan < (uint32_t)kLongSpan ? total - k * kLongSpan : (uint32_t)kLongSpan;
        work[at + 2 * k] = (uint32_t)p + k * kLongSpan;
        work[at + 2 * k + 1] = span;
    }
}

// One wave per work item (first position, count <= kLongSpan): its 64 / LPR lane groups sum every (64 / LPR)-th staging
// row of the span, the groups' sums meet through shuffles, the first group adds the result into the gradient row.
template <int LPR>
__global__ __launch_bounds__(256) void k_seg_sum(const uint32_t *__restrict__ work, const uint32_t *__restrict__ n_work,
                                                 int n_users, const uint32_t *__restrict__ sk, const uint32_t *__restrict__ sv,
                                                 const float *__restrict__ stage, float *gU, float *gI) {
    constexpr int d = 4 * LPR, G = kWave / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
    const uint32_t nw = *n_work >> 1;
    for (uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6); e < nw; e += gridDim.x * 4) {
        const uint32_t p = work[2 * e], len = work[2 * e + 1];
        const uint32_t key = sk[p];
        const float *src = stage + 4 * sub;
        float4 acc = make_float4(0, 0, 0, 0);
        uint32_t k = grp;
        auto row_of = [&](uint32_t q) { return sv ? sv[q] : q; };       // (sv == NULL: the staging rows are in list order)
        for (; k + 3 * G < len; k += 4 * G) {
            const uint32_t r0 = row_of(p + k), r1 = row_of(p + k + G), r2 = row_of(p + k + 2 * G), r3 = row_of(p + k + 3 * G);
            const float4 x0 = ld4(src + (size_t)r0 * d), x1 = ld4(src + (size_t)r1 * d);
            const float4 x2 = ld4(src + (size_t)r2 * d), x3 = ld4(src + (size_t)r3 * d);
            acc = add4(add4(add4(add4(acc, x0), x1), x2), x3);
        }
        for (; k < len; k += G) acc = add4(acc, ld4(src + (size_t)row_of(p + k) * d));
#pragma unroll
        for (int m = LPR; m < kWave; m <<= 1) {
            acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
            acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
        }
        if (grp == 0) {
            const bool is_user = key < (uint32_t)n_users;
            float *dst = (is_user ? gU : gI) + (size_t)(is_user ? key : key - (uint32_t)n_users) * d + 4 * sub;
            MACR_ATOMIC_ADD(dst + 0, acc.x); MACR_ATOMIC_ADD(dst + 1, acc.y);
            MACR_ATOMIC_ADD(dst + 2, acc.z); MACR_ATOMIC_ADD(dst + 3, acc.w);
        }
    }
}

// ----------------------------------------------------------------------------
// reg_scatter (LightGCN): the l2 regulariser acts on the EGO rows
// (macr_lightgcn/LightGCN.py:525-528): G[row] += (decay/batch_size)*T[row] for the
// batch rows, and the sum of squares for emb_loss.  One wave per triple.
// ----------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_reg_scatter(int B, int item_off, const int32_t *__restrict__ u,
                                                     const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                     const float *__restrict__ T, float *G, float coef,
                                                     float *__restrict__ part) {
    // u, i, j: the batch grouped by positive item when the caller has it (batch_bucket_block): a block owns kChunkT
    // consecutive slots and adds equal positive rows of its chunk once (combine_positive_rows) -- the hot item of a
    // batch is referenced hundreds of times, and that many atomics on one row serialise (19.6 -> see DESIGN.md)
    constexpr int EPL = WaveRow<D>::EPL;
    __shared__ float red[16];
    __shared__ float s_gi[kChunkT][D];
    __shared__ int s_pos[kChunkT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool act = lane < WaveRow<D>::kActive;
    float sq = 0.f;
    for (int chunk = blockIdx.x; chunk * kChunkT < B; chunk += gridDim.x) {
#pragma unroll
        for (int q = 0; q < kChunkT / 4; ++q) {
            const int slot = wid * (kChunkT / 4) + q, t = chunk * kChunkT + slot;
            if (t >= B) { if (lane == 0) s_pos[slot] = -1; continue; }
            const int ru = u[t], ri = i[t] + item_off, rj = j[t] + item_off;
            if (act) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int k = lane + 64 * e;
                    const float xu = T[(size_t)ru * D + k], xi = T[(size_t)ri * D + k], xj = T[(size_t)rj * D + k];
                    sq = fmaf(xu, xu, fmaf(xi, xi, fmaf(xj, xj, sq)));
                    if (G) {
                        MACR_ATOMIC_ADD(G + (size_t)ru * D + k, coef * xu);
                        MACR_ATOMIC_ADD(G + (size_t)rj * D + k, coef * xj);
                        s_gi[slot][k] = coef * xi;
                    }
                }
            }
            if (lane == 0) s_pos[slot] = ri;
        }
        __syncthreads();
        if (G) combine_positive_rows<D>(s_pos, s_gi, G);
        __syncthreads();
    }
    const float s0 = block_sum(sq, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * kPartStride] = s0;    // slot 0 only
}

}  // namespace macr

// ============================================================================
// Host side: workspace carving + launch sequences (C ABI)
// ============================================================================
namespace macr {

struct SparseCtx {                // = spmm_kernels.hip
    int32_t *cnt;                 // [N] references of the current batch per row (0: not a row of the batch)
    const int32_t *u, *i, *j;     // the batch: rows u[b], n_users + i[b], n_users + j[b]
    int B, n_users;
    int chunk;                    // rows longer than this are hub rows (cut into pieces by the plan); INT_MAX without a plan
    int count;                    // kSparseOut: 1 = count the references into cnt
};
struct AdamFuse {                 // = spmm_kernels.hip: dense Adam on the ego table in the epilogue of the last backward layer
    float *T, *m, *v;
    const StepScalars *scal;
    float b1, b2, eps, coef;
    float *dE;
    double *emb_acc;
};
int launch_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col, const float *val,
                     const void *plan_dev, const void *plan_host_header, const float *E0, float *E, float *work,
                     hipStream_t st, const SparseCtx *sp, int sparse_mode, const AdamFuse *fuse);   // spmm_kernels.hip
constexpr int kSparseOut = 1, kSparseIn = 2;                                          // = spmm_kernels.hip
constexpr int kEmbSlots = 2048;

// The tail of a fused LightGCN step: Adam on the branch vectors (their gradients are the partial rows pair_bwd left) and
// the losses -- emb_loss from the sums of cnt * |T row|^2 the fused epilogue accumulated (cleared here for the next step).
__global__ __launch_bounds__(256) void k_lgcn_finalize(AdamArgs a, const StepScalars *scal, LossArgs L, double *emb_acc) {
    __shared__ float4 s_red[256];
    if (blockIdx.x < a.n_seg) { adam_block<true>(a, a.seg[blockIdx.x].first_block, scal->lr_t, s_red); return; }
    // the whole block reads the slots (all loads in flight at once), one wave finishes
    __shared__ double s_sq[4];
    double sq = 0.0;
    double part[kEmbSlots / 256];
#pragma unroll
    for (int k = 0; k < kEmbSlots / 256; ++k) part[k] = emb_acc[k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < kEmbSlots / 256; ++k) { sq += part[k]; emb_acc[k * 256 + threadIdx.x] = 0.0; }
    sq = wave_sum_d(sq);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x < 64) finalize_losses(L, threadIdx.x, s_sq[0] + s_sq[1] + s_sq[2] + s_sq[3]);
}

#define MACR_DISPATCH_D(d, ...)                                  \
    switch (d) {                                                 \
        case 32:  { constexpr int D = 32;  __VA_ARGS__; } break; \
        case 64:  { constexpr int D = 64;  __VA_ARGS__; } break; \
        case 128: { constexpr int D = 128; __VA_ARGS__; } break; \
        case 256: { constexpr int D = 256; __VA_ARGS__; } break; \
    }

static inline int bxb_rows(int B) {
    // rows per lane of the bxb kernel: fewer for small B so the chip stays full
#ifdef MACR_BXB_ROWS
    return MACR_BXB_ROWS;
#endif
    static const int forced = getenv("MACR_BXB_ROWS") ? atoi(getenv("MACR_BXB_ROWS")) : 0;     // A/B switch: 1, 2 or 4
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    if (B >= 4096) return 4;            // (measured, same box: 34.5 against 35.8 us per step at B = 4096 with 2; 8 rows at B = 8192: 79 against 76)
    return 1;
}

static inline bool use_staging(int B) {
    // MACR_STAGING=1/0 forces the large-batch / small-batch gradient path (measurements); default: by batch size
    static const char *env = getenv("MACR_STAGING");
    if (env && (env[0] == '0' || env[0] == '1')) return env[0] == '1';
    return B > kSmallBatchMax;
}

struct PairWs {
    StepScalars *scal;
    float *gw;          // [kBranchSlots][2*d] partial rows of the branch-vector gradients
    bool staged;        // large-batch path: staging + sorted references + segment reduce
    int32_t *perm, *us, *is, *js;            // small path: the batch grouped by positive item  [B] each
    uint32_t *ska, *sva, *skb, *svb;         // staged: sort buffers [3B] each
    uint32_t *ghist;                         // staged: [sort_hist_words(3B)]
    uint32_t *n_work;                        // staged: length of k_seg_scan's work list
    float *stage;                            // staged: [3][B][d] gradient rows in batch order
    int nblk_bwd;
    float *fwd;         // [7*Bp]
    float *part;        // [nblk_pair*4]
    float *part2;       // [nblk_pair*4]   (LightGCN ego regulariser)
    float *lpart;       // [nrb*ncb]
    float *rowpart;     // [ncb*2*Bp]
    float *colpart;     // [nrb*2*Bp]
    int Bp, nrb, ncb, rows, nblk_pair;
    int nneu, ncbx;     // neutral blocks per row block of the (B,B) launch (0: off); ncbx = ncb + nneu row-sum slabs / loss partials per row block
    size_t bytes;
};

static PairWs carve_pair_ws(void *base, int B, int d, bool force_staged = false) {
    PairWs w;
    const int lpr = d / 4, rpb = 256 / lpr;
    w.Bp = (int)align_up((size_t)B, 256);
    w.rows = bxb_rows(B);
    w.nrb = w.Bp / (64 * w.rows);
    w.ncb = w.Bp / 256;
    static const bool no_neutral = getenv("MACR_BXB_NEUTRAL") && getenv("MACR_BXB_NEUTRAL")[0] == '0';      // A/B switch
    w.nneu = (!force_staged && B % 256 == 0 && !no_neutral) ? 1 : 0;     // (full batches; the row-sharded step keeps the plain launch)
    w.ncbx = w.ncb + w.nneu;
    w.nblk_pair = (B + rpb - 1) / rpb;
    w.staged = force_staged || use_staging(B);
    if (w.staged) w.nblk_bwd = w.nblk_pair < 4096 ? w.nblk_pair : 4096;                      // grid-strided row groups
    else w.nblk_bwd = (B + kChunkT - 1) / kChunkT < 1024 ? (B + kChunkT - 1) / kChunkT : 1024;   // 16 consecutive slots per block pass
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { void *r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
    w.scal = static_cast<StepScalars *>(take(sizeof(StepScalars)));
    w.gw = static_cast<float *>(take((size_t)kBranchSlots * 2 * d * 4));
    w.fwd = static_cast<float *>(take((size_t)7 * w.Bp * 4));
    // (the loss-only normalbce pass launches min(ceil(B / kChunkT), 1024) blocks whatever path the workspace is carved for)
    const int nblk_loss = (B + kChunkT - 1) / kChunkT < 1024 ? (B + kChunkT - 1) / kChunkT : 1024;
    int npart = w.nblk_pair > w.nblk_bwd ? w.nblk_pair : w.nblk_bwd;
    npart = npart > nblk_loss ? npart : nblk_loss;
    w.part = static_cast<float *>(take((size_t)npart * kPartStride * 4));
    w.part2 = static_cast<float *>(take((size_t)npart * kPartStride * 4));
    w.lpart = static_cast<float *>(take((size_t)w.nrb * w.ncbx * 4));
    w.rowpart = static_cast<float *>(take((size_t)w.ncbx * 2 * w.Bp * 4));
    w.colpart = static_cast<float *>(take((size_t)w.nrb * 2 * w.Bp * 4));
    const size_t nsort = 3 * (size_t)B;
    w.ska = w.sva = w.skb = w.svb = nullptr;
    w.perm = w.us = w.is = w.js = nullptr; w.ghist = nullptr; w.stage = nullptr; w.n_work = nullptr;
    if (w.staged) {
        w.ska = static_cast<uint32_t *>(take(nsort * 4)); w.sva = static_cast<uint32_t *>(take(nsort * 4));
        w.skb = static_cast<uint32_t *>(take(nsort * 4)); w.svb = static_cast<uint32_t *>(take(nsort * 4));
        w.ghist = static_cast<uint32_t *>(take(sort_hist_words((int)nsort) * 4));
        w.n_work = static_cast<uint32_t *>(take(4));
        w.stage = static_cast<float *>(take(nsort * d * 4));
    } else {
        w.perm = static_cast<int32_t *>(take((size_t)B * 4)); w.us = static_cast<int32_t *>(take((size_t)B * 4));
        w.is = static_cast<int32_t *>(take((size_t)B * 4));   w.js = static_cast<int32_t *>(take((size_t)B * 4));
    }
    w.bytes = off;
    return w;
}

// grid: nbxb (B,B) blocks, then the blocks that group the batch (small path), then the pending Adam blocks
template <int R>
static void launch_bxb_rows(const PairWs &ws, int B, const AdamArgs *pending, long long n_adam_blocks,
                            const BatchSort &sort, hipStream_t st) {
    const int nbxb = ws.ncb * ws.nrb, nsort = (sort.B + kBucketSpan - 1) / kBucketSpan + sort.nneu * ws.nrb;    // (+ the neutral blocks)
    const bool full = B % 256 == 0;
#ifdef MACR_ABL_XNOADAM
    n_adam_blocks = 0;                 // timing probe: the ADAM instantiation of the kernel without any Adam block (wrong results)
#endif
    if (pending && pending->lazy) {
        const unsigned grid = (unsigned)(nbxb + nsort + n_adam_blocks);
        if (full) k_bxb<R, true, 2><<<grid, 256, 0, st>>>(B, ws.Bp, ws.ncb, nbxb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, *pending, ws.scal, sort);
        else      k_bxb<R, false, 2><<<grid, 256, 0, st>>>(B, ws.Bp, ws.ncb, nbxb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, *pending, ws.scal, sort);
    } else if (pending) {
        const unsigned grid = (unsigned)(nbxb + nsort + n_adam_blocks);
        if (full) k_bxb<R, true, 1><<<grid, 256, 0, st>>>(B, ws.Bp, ws.ncb, nbxb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, *pending, ws.scal, sort);
        else      k_bxb<R, false, 1><<<grid, 256, 0, st>>>(B, ws.Bp, ws.ncb, nbxb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, *pending, ws.scal, sort);
    } else {
        AdamArgs none;
        none.n_seg = 0;
        if (full) k_bxb<R, true, 0><<<nbxb + nsort, 256, 0, st>>>(B, ws.Bp, ws.ncb, nbxb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, none, ws.scal, sort);
        else      k_bxb<R, false, 0><<<nbxb + nsort, 256, 0, st>>>(B, ws.Bp, ws.ncb, nbxb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, none, ws.scal, sort);
    }
}

static BatchSort batch_sort_args(const PairWs &ws, int B, const int32_t *u, const int32_t *i, const int32_t *j) {
    BatchSort s;
    s.u = u; s.i = i; s.j = j; s.B = B; s.rb0 = 0;
    s.perm = ws.perm; s.us = ws.us; s.is = ws.is; s.js = ws.js;
    s.outflag = nullptr; s.nneu = 0; s.ncbx = ws.ncb;
    return s;
}

// staged path: sort the 3B references by row, then one owner per row sums its staging rows into gU/gI
// sv_sorted != NULL: INDEX mode -- rows whose references lie in one chunk get a flag naming them instead of a sum
// (*sv_sorted = the sorted list's values, for the indexed Adam pass that must follow)
struct RefSort { const uint32_t *sk, *sv; uint32_t *free_key, *free_val; };
static RefSort launch_ref_sort(int B, int n_urows, int n_irows, const int32_t *u, const int32_t *i, const int32_t *j,
                               const PairWs &ws, hipStream_t st) {
    const int n = 3 * B;
    k_refs_init<<<(B + 255) / 256 < 2048 ? (B + 255) / 256 : 2048, 256, 0, st>>>(B, n_urows, u, i, j, ws.ska, ws.sva, ws.n_work);
    const int flip = launch_radix_sort(ws.ska, ws.sva, ws.skb, ws.svb, n, (uint32_t)(n_urows + n_irows - 1), ws.ghist, st);
    RefSort r;
    r.sk = flip ? ws.skb : ws.ska; r.sv = flip ? ws.svb : ws.sva;
    r.free_key = flip ? ws.ska : ws.skb; r.free_val = flip ? ws.sva : ws.svb;     // the buffers the sort no longer needs
    return r;
}
// place[reference] = its position in the sorted list (the inverse of the list's values)
__global__ __launch_bounds__(256) void k_ref_place(int n, const uint32_t *__restrict__ sv, uint32_t *__restrict__ place) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) place[sv[p]] = (uint32_t)p;
}
// INDEX mode behind a sort: flags that name each row's run, the long runs summed into gU/gI.  listed: the staging rows
// are in list order (sv not needed to find them)
static int launch_seg_index(int B, int d, int n_urows, int n_irows, const RefSort &r, bool listed, float *gU, float *gI,
                            int32_t *tU, int32_t *tI, const PairWs &ws, hipStream_t st) {
    const int n = 3 * B;
    k_seg_scan<<<(n + 255) / 256, 256, 0, st>>>(n, n_urows, (uint32_t)(n_urows + n_irows), r.sk, tU, tI, r.free_key, ws.n_work);
    MACR_CHECK_LAUNCH("seg_index", st);
    MACR_DISPATCH_LPR(d, (k_seg_sum<LPR><<<1024, 256, 0, st>>>(r.free_key, ws.n_work, n_urows, r.sk, listed ? nullptr : r.sv, ws.stage, gU, gI)));
    MACR_CHECK_LAUNCH("seg_sum", st);
    return MACR_OK;
}
static int launch_ref_sort_reduce(int B, int d, int n_urows, int n_irows, const int32_t *u, const int32_t *i,
                                  const int32_t *j, float *gU, float *gI, int32_t *tU, int32_t *tI, const PairWs &ws,
                                  hipStream_t st, const uint32_t **sv_sorted = nullptr) {
    const int n = 3 * B;
    const RefSort r = launch_ref_sort(B, n_urows, n_irows, u, i, j, ws, st);
    MACR_CHECK_LAUNCH("ref_sort", st);
    if (sv_sorted) {
        *sv_sorted = r.sv;
        return launch_seg_index(B, d, n_urows, n_irows, r, false, gU, gI, tU, tI, ws, st);
    }
    MACR_DISPATCH_LPR(d, (k_seg_reduce<LPR><<<(n + 256 / LPR - 1) / (256 / LPR), 256, 0, st>>>(
                             n, n_urows, (uint32_t)(n_urows + n_irows), r.sk, r.sv, ws.stage, gU, gI, tU, tI)));
    MACR_CHECK_LAUNCH("seg_reduce", st);
    return MACR_OK;
}

// forward + (B,B) + backward of the pair loss; on return (stream order) gU/gI hold the de-duplicated gradient rows.
// n_urows / n_irows: rows of the tables Usrc / Isrc (sort key range).
static int launch_pair(int kind, int B, int d, int n_urows, int n_irows, const int32_t *u, const int32_t *i,
                       const int32_t *j, const float *Usrc, const float *Isrc, const float *w, const float *wu,
                       float *gU, float *gI, int32_t *tU, int32_t *tI, float coef, int reg_on_gathered,
                       float *adam_pow, const macr_hyper *hp, const PairWs &ws, hipStream_t st,
                       const PendingAdam *pa = nullptr, const AdamArgs *pending = nullptr,
                       long long n_pending_blocks = 0, const LossArgs *finalize = nullptr, bool loss_only = false,
                       int32_t *cnt_pos = nullptr, const uint32_t **sv_sorted = nullptr, LazyState *tick = nullptr) {
    const int grid = ws.nblk_pair;
    const int user_branch = kind == MACR_LOSS_RUBIBCEBOTH;
    BatchSort sort = batch_sort_args(ws, B, u, i, j);
    if (ws.staged || loss_only) sort.B = 0;
    sort.outflag = ws.fwd + 6 * (size_t)ws.Bp; sort.nneu = ws.nneu; sort.ncbx = ws.ncbx;      // neutralised columns of the (B,B) launch (pair_fwd flags them)
    if (kind == MACR_LOSS_NORMALBCE && loss_only) {            // forward of the per-pair loss, nothing written but partials
        const int nb = (B + kChunkT - 1) / kChunkT < 1024 ? (B + kChunkT - 1) / kChunkT : 1024;
        MACR_DISPATCH_D(d, (k_pair_normal<D><<<nb, 256, 0, st>>>(B, u, i, j, Usrc, Isrc, nullptr, nullptr, nullptr, nullptr,
                                                                ws.part, coef, reg_on_gathered, adam_pow, nullptr,
                                                                ws.scal, hp->lr, hp->beta1, hp->beta2, nullptr)));
        MACR_CHECK_LAUNCH("pair_normal", st);
        return MACR_OK;
    }
    // Large batch whose Adam pass follows at once (sv_sorted): the references are sorted FIRST -- the order depends on the
    // indices only -- and every gradient row is staged at its reference's position in the sorted list, so that the pass
    // reads a row's <= 16 staging rows as one run (805 MB of scattered 256-byte reads at B = 2^20 became sequential) and
    // needs no list to find them.  MACR_STAGE_LISTED=0: staging in batch order, found through the list (before round 4).
    static const bool listed_ok = !(getenv("MACR_STAGE_LISTED") && atoi(getenv("MACR_STAGE_LISTED")) == 0);
    const bool listed = ws.staged && sv_sorted && listed_ok && !loss_only;
    RefSort rs = {};
    if (listed) {
        rs = launch_ref_sort(B, n_urows, n_irows, u, i, j, ws, st);
        MACR_CHECK_LAUNCH("ref_sort", st);
        k_ref_place<<<(3 * B + 255) / 256 < 4096 ? (3 * B + 255) / 256 : 4096, 256, 0, st>>>(3 * B, rs.sv, rs.free_val);
        MACR_CHECK_LAUNCH("ref_place", st);
        *sv_sorted = nullptr;                                    // (the indexed pass finds the rows by position)
    }
    if (kind == MACR_LOSS_NORMALBCE) {
        if (ws.staged) {
            MACR_DISPATCH_LPR(d, (k_pair_normal_stage<LPR><<<ws.nblk_bwd, 256, 0, st>>>(
                                     B, u, i, j, Usrc, Isrc, ws.stage, ws.part, coef, reg_on_gathered, adam_pow, adam_pow,
                                     ws.scal, hp->lr, hp->beta1, hp->beta2, listed ? rs.free_val : nullptr)));
            MACR_CHECK_LAUNCH("pair_normal", st);
            if (listed) return launch_seg_index(B, d, n_urows, n_irows, rs, true, gU, gI, tU, tI, ws, st);
            return launch_ref_sort_reduce(B, d, n_urows, n_irows, u, i, j, gU, gI, tU, tI, ws, st, sv_sorted);
        }
        k_batch_sort<<<(B + kBucketSpan - 1) / kBucketSpan, 256, 0, st>>>(sort);
        MACR_CHECK_LAUNCH("batch_sort", st);
        MACR_DISPATCH_D(d, (k_pair_normal<D><<<ws.nblk_bwd, 256, 0, st>>>(B, ws.us, ws.is, ws.js, Usrc, Isrc, gU, gI, tU, tI,
                                                                      ws.part, coef, reg_on_gathered, adam_pow, adam_pow,
                                                                      ws.scal, hp->lr, hp->beta1, hp->beta2, cnt_pos)));
        MACR_CHECK_LAUNCH("pair_normal", st);
        return MACR_OK;
    }
    if (pa && pa->lazy) {
        MACR_DISPATCH_LPR(d, (k_pair_fwd<LPR, 2><<<grid, 256, 0, st>>>(B, ws.Bp, u, i, j, Usrc, Isrc, w, wu, ws.fwd,
                                                                      ws.part, reg_on_gathered, ws.gw, *pa, user_branch)));
    } else if (pa) {
        MACR_DISPATCH_LPR(d, (k_pair_fwd<LPR, 1><<<grid, 256, 0, st>>>(B, ws.Bp, u, i, j, Usrc, Isrc, w, wu, ws.fwd,
                                                                      ws.part, reg_on_gathered, ws.gw, *pa, user_branch)));
    } else {
        PendingAdam none = {};
        MACR_DISPATCH_LPR(d, (k_pair_fwd<LPR, 0><<<grid, 256, 0, st>>>(B, ws.Bp, u, i, j, Usrc, Isrc, w, wu, ws.fwd,
                                                                      ws.part, reg_on_gathered, ws.gw, none, user_branch)));
    }
    MACR_CHECK_LAUNCH("pair_fwd", st);
    switch (ws.rows) {
        case 1: launch_bxb_rows<1>(ws, B, pending, n_pending_blocks, sort, st); break;
        case 2: launch_bxb_rows<2>(ws, B, pending, n_pending_blocks, sort, st); break;
        default: launch_bxb_rows<4>(ws, B, pending, n_pending_blocks, sort, st); break;
    }
    MACR_CHECK_LAUNCH(pending ? "bxb+adam" : "bxb", st);
    if (loss_only) return MACR_OK;
    LossArgs L;
    if (finalize) L = *finalize; else L.losses = nullptr;
    if (ws.staged) {
        MACR_DISPATCH_LPR(d, (k_pair_bwd_stage<LPR><<<ws.nblk_bwd + 1, 256, 0, st>>>(
                                 B, ws.Bp, ws.nrb, ws.ncb, u, i, j, Usrc, Isrc, w, wu, ws.fwd, ws.rowpart, ws.colpart, ws.stage,
                                 ws.gw, hp->alpha, hp->beta, coef, adam_pow, ws.scal, hp->lr, hp->beta1, hp->beta2, L,
                                 listed ? rs.free_val : nullptr, 0, tick, ws.nneu)));
        MACR_CHECK_LAUNCH("pair_bwd", st);
        if (listed) return launch_seg_index(B, d, n_urows, n_irows, rs, true, gU, gI, tU, tI, ws, st);
        return launch_ref_sort_reduce(B, d, n_urows, n_irows, u, i, j, gU, gI, tU, tI, ws, st, sv_sorted);
    }
    MACR_DISPATCH_D(d, (k_pair_bwd<D><<<ws.nblk_bwd + 1, 256, 0, st>>>(B, ws.Bp, ws.nrb, ws.ncb, ws.perm, ws.us, ws.is, ws.js,
                                                                      Usrc, Isrc, w, wu, ws.fwd, ws.rowpart, ws.colpart, gU,
                                                                      gI, tU, tI, ws.gw, hp->alpha, hp->beta, coef, adam_pow,
                                                                      ws.scal, hp->lr, hp->beta1, hp->beta2, L, cnt_pos, tick, ws.nneu)));
    MACR_CHECK_LAUNCH("pair_bwd", st);
    return MACR_OK;
}

static void add_seg(AdamArgs &a, float *theta, float *m, float *v, float *g, int32_t *touched, long long rows,
                    long long &next_block, int n_parts = 0, int part_stride = 0) {
    AdamSeg &s = a.seg[a.n_seg++];
    s.theta = theta; s.m = m; s.v = v; s.g = g; s.touched = touched; s.stamp = nullptr;
    s.sv = nullptr; s.stage = nullptr;
    s.n_parts = n_parts; s.part_stride = part_stride;
    s.n_vec = rows * a.lpr;
    s.first_block = next_block;
    next_block += (s.n_vec + kAdamVecPerBlock - 1) / kAdamVecPerBlock;
}

static int validate_lazy(const macr_lazy_adam *lz, bool need_p, bool need_q, const char *who);
static int validate_hyper(const macr_hyper *hp, const char *who) {
    MACR_REQUIRE(hp, MACR_E_INVALID, "%s: hyper is null", who);
    MACR_REQUIRE(hp->batch_size_cfg > 0, MACR_E_INVALID, "%s: batch_size_cfg=%d", who, hp->batch_size_cfg);
    MACR_REQUIRE(hp->beta1 > 0.f && hp->beta1 < 1.f && hp->beta2 > 0.f && hp->beta2 < 1.f, MACR_E_INVALID,
                 "%s: adam betas (%g,%g) outside (0,1)", who, hp->beta1, hp->beta2);
    return MACR_OK;
}

}  // namespace macr

using namespace macr;

extern "C" size_t macr_mf_train_workspace_bytes(int B, int d) {
    if (B <= 0 || !dim_supported(d)) return 0;
    return carve_pair_ws(nullptr, B, d).bytes;
}

namespace macr {
// Adam segment lists of the MF model.  tables: P and Q (row-flag protocol); branch vectors w, w_user (partial rows
// of pair_bwd) as the loss kind trains them (model.py:74 / :69 / :95).
static void mf_adam_args(AdamArgs &a, long long &nb, bool tables, int loss_kind, int d, int n_users, int n_items,
                         float *P, float *Q, float *w, float *wu, float *mP, float *vP, float *mQ, float *vQ,
                         float *mw, float *vw, float *mwu, float *vwu, float *gP, float *gQ, int32_t *tP,
                         int32_t *tQ, const macr_hyper *hp, const PairWs &ws) {
    a.n_seg = 0;
    a.lpr = d / 4; a.lpr_shift = d == 32 ? 3 : d == 64 ? 4 : d == 128 ? 5 : 6;
    a.b1 = hp->beta1; a.b2 = hp->beta2; a.eps = hp->adam_eps;
    nb = 0;
    if (tables) {
        add_seg(a, P, mP, vP, gP, tP, n_users, nb);
        add_seg(a, Q, mQ, vQ, gQ, tQ, n_items, nb);
    }
    // w: both branch losses; w_user: rubibceboth only (its gradient is None elsewhere -> TF leaves it alone)
    if (loss_kind != MACR_LOSS_NORMALBCE) add_seg(a, w, mw, vw, ws.gw, nullptr, 1, nb, kBranchSlots, 2 * d);
    if (loss_kind == MACR_LOSS_RUBIBCEBOTH) add_seg(a, wu, mwu, vwu, ws.gw + d, nullptr, 1, nb, kBranchSlots, 2 * d);
}
}  // namespace macr

namespace macr {
// lazy dense Adam on the tables of the MF model: stamps of the segments, the ring, the period
static void lazy_adam_args(AdamArgs &a, const macr_lazy_adam *lz, const float *P) {
    a.lazy = static_cast<const LazyState *>(lz->state);
    a.period = lz->period;
    for (int k = 0; k < a.n_seg; ++k)
        if (a.seg[k].n_parts == 0) a.seg[k].stamp = a.seg[k].theta == P ? lz->stampP : lz->stampQ;
}
// every row of both tables to the current step (k_adam_lazy in flush form)
static int lazy_flush_tables(int d, long long n_rows_p, long long n_rows_q, float *P, float *Q, float *mP, float *vP, float *mQ,
                             float *vQ, const macr_hyper *hp, const macr_lazy_adam *lazy, hipStream_t st) {
    AdamArgs a;
    long long nb = 0;
    a.n_seg = 0; a.lpr = d / 4; a.lpr_shift = d == 32 ? 3 : d == 64 ? 4 : d == 128 ? 5 : 6;
    a.b1 = hp->beta1; a.b2 = hp->beta2; a.eps = hp->adam_eps;
    if (n_rows_p) { add_seg(a, P, mP, vP, nullptr, nullptr, n_rows_p, nb); a.seg[a.n_seg - 1].stamp = lazy->stampP; }
    if (n_rows_q) { add_seg(a, Q, mQ, vQ, nullptr, nullptr, n_rows_q, nb); a.seg[a.n_seg - 1].stamp = lazy->stampQ; }
    if (a.n_seg == 0) return MACR_OK;
    LazyArgs z;
    z.state = static_cast<const LazyState *>(lazy->state); z.period = 1; z.n_tab = a.n_seg;
    for (int k = 0; k < 3; ++k) z.sweep_first[k] = k < a.n_seg ? a.seg[k].first_block : nb;
    z.touch_first = z.branch_first = nb;
    z.n_refs = 0; z.n_users = 0; z.item_seg = 0; z.key_end = 0; z.sk = nullptr;
    k_adam_lazy<false, false><<<(unsigned)nb, 256, 0, st>>>(a, z, nullptr);
    MACR_CHECK_LAUNCH("lazy_flush", st);
    return MACR_OK;
}

static int mf_train_step(int loss_kind, int B, int d, int n_users, int n_items, const int32_t *u,
                         const int32_t *i, const int32_t *j, float *P, float *Q, float *w, float *wu,
                         float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                         float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                         float *adam_pow, const macr_hyper *hp, float *losses, int flags, const macr_lazy_adam *lz,
                         void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_NORMALBCE || loss_kind == MACR_LOSS_RUBIBCEBOTH || loss_kind == MACR_LOSS_RUBIBCE,
                 MACR_E_INVALID, "mf_train_step: loss_kind=%d", loss_kind);
    MACR_REQUIRE(B > 0 && n_users > 0 && n_items > 0, MACR_E_INVALID, "mf_train_step: B=%d n_users=%d n_items=%d", B,
                 n_users, n_items);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "mf_train_step: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(u && i && j && P && Q && mP && vP && mQ && vQ && gP && gQ && touchedP && touchedQ && adam_pow &&
                     losses && workspace, MACR_E_INVALID, "mf_train_step: null pointer");
    MACR_REQUIRE(loss_kind == MACR_LOSS_NORMALBCE || (w && wu && mw && vw && mwu && vwu), MACR_E_INVALID,
                 "mf_train_step: rubibceboth needs w, wu and their Adam slots");
    MACR_REQUIRE((flags & ~(MACR_STEP_DEFER | MACR_STEP_PENDING)) == 0, MACR_E_INVALID, "mf_train_step: flags=%d", flags);
    MACR_REQUIRE(!flags || loss_kind != MACR_LOSS_NORMALBCE, MACR_E_INVALID,
                 "mf_train_step: deferred mode exists for the (B,B) losses only (flags=%d)", flags);
    MACR_REQUIRE(!lz || loss_kind != MACR_LOSS_NORMALBCE, MACR_E_UNSUPPORTED,
                 "mf_train_step_lazy: the lazy pass rides in the (B,B) launch (rubibceboth, rubibce)");
    if (int e = validate_hyper(hp, "mf_train_step")) return e;
    if (lz) if (int e = validate_lazy(lz, true, true, "mf_train_step_lazy")) return e;
    PairWs ws = carve_pair_ws(workspace, B, d);
    MACR_REQUIRE(workspace_bytes >= ws.bytes, MACR_E_WORKSPACE, "mf_train_step: workspace %zu < %zu bytes",
                 workspace_bytes, ws.bytes);
    MACR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID,
                 "mf_train_step: workspace must be 256-byte aligned");
    hipStream_t st = as_stream(stream);
    const float coef = hp->decay / (float)hp->batch_size_cfg;       // d reg / d row  (model.py:219-221)
    const bool rubi = loss_kind != MACR_LOSS_NORMALBCE;          // the (B,B) losses: rubibceboth, rubibce
    LossArgs L;
    L.part = ws.part; L.n_part = rubi ? ws.nblk_pair : ws.nblk_bwd;
    L.part2 = nullptr; L.n_part2 = 0;
    L.lpart = ws.lpart; L.n_lpart = rubi ? ws.nrb * ws.ncbx : 0;
    L.kind = loss_kind; L.B = B; L.batch_size_cfg = hp->batch_size_cfg;
    L.alpha = hp->alpha; L.beta = hp->beta; L.decay = hp->decay; L.losses = losses;
    AdamArgs a;
    long long nb = 0;
    const bool pending = flags & MACR_STEP_PENDING, defer = flags & MACR_STEP_DEFER;
    PendingAdam pa = {};
    if (pending) {
        // the previous call left its dense pass pending: pair_fwd looks one update ahead, the pass itself (tables
        // and branch vectors) rides in the bxb launch
        pa.mU = mP; pa.vU = vP; pa.gU = gP; pa.mI = mQ; pa.vI = vQ; pa.gI = gQ; pa.tU = touchedP; pa.tI = touchedQ;
        pa.mw = mw; pa.vw = vw; pa.mwu = mwu; pa.vwu = vwu; pa.scal = ws.scal;
        pa.b1 = hp->beta1; pa.b2 = hp->beta2; pa.eps = hp->adam_eps;
        mf_adam_args(a, nb, true, loss_kind, d, n_users, n_items, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ,
                     touchedP, touchedQ, hp, ws);
        if (lz) {
            pa.stU = lz->stampP; pa.stI = lz->stampQ; pa.lazy = static_cast<const LazyState *>(lz->state);
            lazy_adam_args(a, lz, P);
        }
    }
    // Large batches, step complete in this call: the reference sort is followed by the Adam pass at once, which can sum
    // a row's staged gradient rows itself instead of reading a row some kernel wrote for it (adam_block INDEXED).
    // (MACR_SEG_UNFUSED=1: the segment reduce writes every row, as in deferred mode -- for A/B measurements and tests.)
    const char *unfused = getenv("MACR_SEG_UNFUSED");
    const bool indexed = ws.staged && !flags && !lz && (size_t)3 * B <= kRefMaxRefs && !(unfused && unfused[0] == '1');
    const uint32_t *sv_sorted = nullptr;
    if (int e = launch_pair(loss_kind, B, d, n_users, n_items, u, i, j, P, Q, w, wu, gP, gQ, touchedP, touchedQ, coef, 1,
                            adam_pow, hp, ws, st, pending ? &pa : nullptr, pending ? &a : nullptr, nb, defer ? &L : nullptr,
                            false, nullptr, indexed ? &sv_sorted : nullptr, lz ? static_cast<LazyState *>(lz->state) : nullptr))
        return e;
    if (defer) return MACR_OK;
    mf_adam_args(a, nb, true, loss_kind, d, n_users, n_items, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ,
                 touchedP, touchedQ, hp, ws);
    if (lz) {
        // the step completes in this call: its own pass over the flagged rows and the step's sweep, then every row to the
        // step -- P, Q and the slots are up to date when the call returns
        lazy_adam_args(a, lz, P);
        k_adam_lazy_scan<<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
        MACR_CHECK_LAUNCH("adam_lazy", st);
        return lazy_flush_tables(d, n_users, n_items, P, Q, mP, vP, mQ, vQ, hp, lz, st);
    }
    if (indexed) {
        a.seg[0].sv = a.seg[1].sv = sv_sorted;
        a.seg[0].stage = a.seg[1].stage = ws.stage;
        k_adam_dense<true><<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
        MACR_CHECK_LAUNCH("adam_indexed", st);
        return MACR_OK;
    }
    k_adam_dense<false><<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
    MACR_CHECK_LAUNCH("adam_dense", st);
    return MACR_OK;
}

static int mf_train_flush(int loss_kind, int B, int d, int n_users, int n_items, float *P, float *Q, float *w, float *wu,
                          float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                          float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                          const macr_hyper *hp, const macr_lazy_adam *lz, void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_RUBIBCEBOTH || loss_kind == MACR_LOSS_RUBIBCE, MACR_E_INVALID,
                 "mf_train_flush: loss_kind=%d has no deferred mode", loss_kind);
    MACR_REQUIRE(B > 0 && n_users > 0 && n_items > 0, MACR_E_INVALID, "mf_train_flush: B=%d n_users=%d n_items=%d", B,
                 n_users, n_items);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "mf_train_flush: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(P && Q && w && wu && mP && vP && mQ && vQ && mw && vw && mwu && vwu && gP && gQ && touchedP &&
                     touchedQ && workspace, MACR_E_INVALID, "mf_train_flush: null pointer");
    if (int e = validate_hyper(hp, "mf_train_flush")) return e;
    if (lz) if (int e = validate_lazy(lz, true, true, "mf_train_flush_lazy")) return e;
    PairWs ws = carve_pair_ws(workspace, B, d);
    MACR_REQUIRE(workspace_bytes >= ws.bytes, MACR_E_WORKSPACE, "mf_train_flush: workspace %zu < %zu bytes",
                 workspace_bytes, ws.bytes);
    AdamArgs a;
    long long nb = 0;
    mf_adam_args(a, nb, true, loss_kind, d, n_users, n_items, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ,
                 touchedP, touchedQ, hp, ws);
    LossArgs L;
    L.losses = nullptr;
    hipStream_t st = as_stream(stream);
    if (lz) {
        lazy_adam_args(a, lz, P);
        k_adam_lazy_scan<<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
        MACR_CHECK_LAUNCH("adam_lazy", st);
        return lazy_flush_tables(d, n_users, n_items, P, Q, mP, vP, mQ, vQ, hp, lz, st);
    }
    k_adam_dense<false><<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
    MACR_CHECK_LAUNCH("adam_dense", st);
    return MACR_OK;
}
}  // namespace macr

extern "C" int macr_mf_train_step(int loss_kind, int B, int d, int n_users, int n_items, const int32_t *u,
                                  const int32_t *i, const int32_t *j, float *P, float *Q, float *w, float *wu,
                                  float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                                  float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                                  float *adam_pow, const macr_hyper *hp, float *losses, int flags, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    return mf_train_step(loss_kind, B, d, n_users, n_items, u, i, j, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ, touchedP,
                         touchedQ, adam_pow, hp, losses, flags, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int macr_mf_train_step_lazy(int loss_kind, int B, int d, int n_users, int n_items, const int32_t *u,
                                       const int32_t *i, const int32_t *j, float *P, float *Q, float *w, float *wu,
                                       float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                                       float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                                       float *adam_pow, const macr_hyper *hp, float *losses, int flags,
                                       const macr_lazy_adam *lazy, void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(lazy, MACR_E_INVALID, "mf_train_step_lazy: lazy is null");
    return mf_train_step(loss_kind, B, d, n_users, n_items, u, i, j, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ, touchedP,
                         touchedQ, adam_pow, hp, losses, flags, lazy, workspace, workspace_bytes, stream);
}

extern "C" int macr_mf_train_flush(int loss_kind, int B, int d, int n_users, int n_items, float *P, float *Q, float *w, float *wu,
                                   float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                                   float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                                   const macr_hyper *hp, void *workspace, size_t workspace_bytes, void *stream) {
    return mf_train_flush(loss_kind, B, d, n_users, n_items, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ, touchedP, touchedQ,
                          hp, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int macr_mf_train_flush_lazy(int loss_kind, int B, int d, int n_users, int n_items, float *P, float *Q, float *w,
                                        float *wu, float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                                        float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ,
                                        const macr_hyper *hp, const macr_lazy_adam *lazy, void *workspace, size_t workspace_bytes,
                                        void *stream) {
    MACR_REQUIRE(lazy, MACR_E_INVALID, "mf_train_flush_lazy: lazy is null");
    return mf_train_flush(loss_kind, B, d, n_users, n_items, P, Q, w, wu, mP, vP, mQ, vQ, mw, vw, mwu, vwu, gP, gQ, touchedP, touchedQ,
                          hp, lazy, workspace, workspace_bytes, stream);
}

// ============================================================================
// Row-sharded training (BASELINE configs[4]: 10 M x 1 M rows, d = 128 over 8 GPUs).
// The reference keeps each table in ONE tf.Variable (macr_mf/model.py:112-113); what does not fit or should not be
// streamed by one GPU is the dense Adam pass -- 24*d bytes of EVERY row per step (33.8 GB at that size).  Rows of P
// and Q, their Adam slots and gradient scratch are therefore range-sharded over the ranks, and a step becomes
//   gather    every rank writes the batch rows it owns into a zero [3][B][d] buffer  -> all-reduce(sum) = every rank
//             holds the batch's 3B rows (exactly one owner per row)
//   forward   per-pair dots and branch factors for the whole batch, on every rank (B*(12d+12) bytes: negligible,
//             deterministic, so no exchange of p, n, a, b is needed)
//   bxb       rank r evaluates row blocks [r*nrb/W, (r+1)*nrb/W) of the (B,B) term into zeroed partial arrays
//             -> all-reduce(sum) of the partials (a few MB) = complete row and column sums everywhere
//   backward  gradient rows of the whole batch into the staging buffer, on every rank (again negligible)
//   apply     each rank sorts the references to ITS rows, segment-reduces them into its gradient shard and runs the
//             dense Adam pass over its shard only: the 24*d*rows bytes per step are divided by W
// The collectives live on the host side (macr_amd/sharded_train.py, torch.distributed = RCCL over xGMI); these entry
// points are the device half.  Results equal the single-GPU step up to summation order.
// ============================================================================
namespace macr {
// Ownership of a table's rows: rank-local row l is global row lo + l * stride, l < n_loc (stride 1: a contiguous range;
// stride = number of ranks, lo = rank: interleaved -- the hot low item ids and the Adam pass spread evenly).
struct Owned { int lo, stride, n_loc; };
__device__ __forceinline__ int owned_local(const Owned o, int row) {      // local index, or -1 when another rank owns the row
    const int rel = row - o.lo;
    if (rel < 0) return -1;
    const int l = o.stride == 1 ? rel : rel / o.stride;
    return (l < o.n_loc && l * o.stride == rel) ? l : -1;
}
template <int LPR, bool LAZY>
__global__ __launch_bounds__(256) void k_rows_gather_owned(int B, const LazyTable P, const Owned ou, const LazyTable Q, const Owned oi,
                                                           const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                                           const int32_t *__restrict__ j, float *__restrict__ rows3,
                                                           const LazyState *state, float b1, float b2, float eps) {
    // LAZY: the tables are updated lazily (k_adam_lazy) -- a row is brought to the current step on its way out
    __shared__ float s_lr[LAZY ? kLazyRing : 1];
    uint32_t T = 0;
    if (LAZY) {
        s_lr[threadIdx.x] = state->lr[threadIdx.x];
        T = state->t;
        __syncthreads();
    }
    const long long n4 = 3LL * B * LPR;
    for (long long g = blockIdx.x * 256LL + threadIdx.x; g < n4; g += gridDim.x * 256LL) {
        const long long ref = g / LPR;
        const int sub = (int)(g % LPR), role = (int)(ref / B), t = (int)(ref % B);
        const int row = role == 0 ? u[t] : role == 1 ? i[t] : j[t];
        const int l = owned_local(role == 0 ? ou : oi, row);
        float4 v = make_float4(0, 0, 0, 0);
        if (l >= 0) {
            if (LAZY) v = lazy_row4<LPR>(role == 0 ? P : Q, l, sub, T, s_lr, b1, b2, eps);
            else v = ld4((role == 0 ? P.theta : Q.theta) + ((size_t)l * LPR + sub) * 4);
        }
        st4(rows3 + (size_t)g * 4, v);
    }
}
__global__ __launch_bounds__(256) void k_iota3(int B, int32_t *__restrict__ a) {
    for (int t = blockIdx.x * 256 + threadIdx.x; t < B; t += gridDim.x * 256) { a[t] = t; a[B + t] = t; a[2 * (size_t)B + t] = B + t; }
}
// sort keys of the references to THIS rank's rows (others: key_end, which k_seg_reduce ignores); value = staging row
__global__ __launch_bounds__(256) void k_shard_keys(int B, const Owned ou, const Owned oi, const int32_t *__restrict__ u,
                                                    const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                    uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                    uint32_t *__restrict__ n_work, LazyState *lazy, const StepScalars *scal) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *n_work = 0u;                                                // work list of k_seg_scan, empty again
        if (lazy) lazy_tick(lazy, scal->lr_t);                       // (the backward kernel of this step wrote lr_t)
    }
    const uint32_t none = (uint32_t)(ou.n_loc + oi.n_loc);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < B; t += gridDim.x * 256) {
        const int ru = owned_local(ou, u[t]), ri = owned_local(oi, i[t]), rj = owned_local(oi, j[t]);
        key[t] = ru >= 0 ? (uint32_t)ru : none;                                   val[t] = (uint32_t)t;
        key[B + t] = ri >= 0 ? (uint32_t)(ou.n_loc + ri) : none;                  val[B + t] = (uint32_t)(B + t);
        key[2 * (size_t)B + t] = rj >= 0 ? (uint32_t)(ou.n_loc + rj) : none;      val[2 * (size_t)B + t] = (uint32_t)(2 * (size_t)B + t);
    }
}
// ---- routing of the split step (macr_shard_route) ----------------------------------------------------------------------------
// References are numbered role * B + t (role 0 / 1 / 2 = user / positive / negative row of position t).  owner(ref) = the rank
// that holds the row (Owned layout of its table), dest(ref) = the rank whose slice [t0, t1) holds position t.  One workgroup:
//   counts[q * W + p]  references owned by q whose position lies in p's slice                       (the all-to-all split sizes)
//   send_ref[0 .. n_send)  references THIS rank owns, ordered by (dest, reference)                  (what it sends, in order)
//   recv_ref[0 .. n_recv)  references of THIS rank's slice, ordered by (owner, reference)           (the order they arrive in)
// Two stable counting sorts over 3B <= 196 608 keys with W <= 16 buckets: thread k owns a contiguous stretch of references,
// counts its stretch per bucket, the per-bucket counts of the 1 024 threads are scanned in LDS, and the thread writes its
// references behind the ones before it -- reference order inside a bucket is preserved by construction.
constexpr int kRouteThreads = 1024, kRouteMaxW = 16;
struct RouteOwner { int stride; const int32_t *bounds; };         // stride >= 2: owner = row % stride; else bounds[W]: first row NOT owned by rank q
__device__ __forceinline__ int route_owner(const RouteOwner o, int W, int row) {
    if (o.bounds == nullptr) return row % o.stride;
    int q = 0;
    while (q + 1 < W && row >= o.bounds[q]) ++q;
    return q;
}
__global__ __launch_bounds__(kRouteThreads) void k_shard_route(int B, int W, int rank, const int32_t *__restrict__ u,
                                                               const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                               const RouteOwner ou, const RouteOwner oi,
                                                               const int32_t *__restrict__ slice_end, int32_t *__restrict__ counts,
                                                               int32_t *__restrict__ send_ref, int32_t *__restrict__ recv_ref) {
    extern __shared__ int32_t s_route[];                           // [2][W][kRouteThreads] per-thread bucket counts, then [W*W] + [2*W] + scratch
    int32_t *cs = s_route, *cr = s_route + W * kRouteThreads, *cnt = cr + W * kRouteThreads, *base = cnt + W * W;
    const int t = threadIdx.x, n = 3 * B;
    const int per = (n + kRouteThreads - 1) / kRouteThreads;
    const int lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    for (int k = t; k < W * W; k += kRouteThreads) cnt[k] = 0;
    for (int q = 0; q < W; ++q) { cs[q * kRouteThreads + t] = 0; cr[q * kRouteThreads + t] = 0; }
    __syncthreads();
    auto keys = [&](int ref, int &owner, int &dest) {
        const int role = ref / B, pos = ref - role * B;
        const int row = role == 0 ? u[pos] : role == 1 ? i[pos] : j[pos];
        owner = route_owner(role == 0 ? ou : oi, W, row);
        dest = 0;
        while (dest + 1 < W && pos >= slice_end[dest]) ++dest;
    };
    for (int ref = lo; ref < hi; ++ref) {
        int owner, dest;
        keys(ref, owner, dest);
        atomicAdd(&cnt[owner * W + dest], 1);
        if (owner == rank) ++cs[dest * kRouteThreads + t];
        if (dest == rank) ++cr[owner * kRouteThreads + t];
    }
    __syncthreads();
    // exclusive scan of every bucket's 1 024 per-thread counts (wave scan, then the 16 wave totals), both sorts
    const int lane = t & 63, wid = t >> 6;
    int32_t *wtot = base + 2 * W;                                  // [2 * W][16] wave totals
    for (int a = 0; a < 2; ++a) {
        int32_t *c = a ? cr : cs;
        for (int q = 0; q < W; ++q) {
            const int v = c[q * kRouteThreads + t];
            int incl = v;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, kWave); if (lane >= m) incl += o; }
            if (lane == 63) wtot[(a * W + q) * 16 + wid] = incl;
            c[q * kRouteThreads + t] = incl - v;                   // exclusive inside the wave
        }
    }
    __syncthreads();
    if (t < 2 * W) {                                               // bucket bases: totals of the buckets before, wave offsets inside a bucket
        int run = 0;
        for (int w2 = 0; w2 < 16; ++w2) { const int x = wtot[t * 16 + w2]; wtot[t * 16 + w2] = run; run += x; }
        base[t] = run;                                             // the bucket's total
    }
    __syncthreads();
    if (t == 0) {
        for (int a = 0; a < 2; ++a) { int run = 0; for (int q = 0; q < W; ++q) { const int x = base[a * W + q]; base[a * W + q] = run; run += x; } }
    }
    __syncthreads();
    int ps[kRouteMaxW], pr[kRouteMaxW];
    for (int q = 0; q < W; ++q) {
        ps[q] = base[q] + wtot[q * 16 + wid] + cs[q * kRouteThreads + t];
        pr[q] = base[W + q] + wtot[(W + q) * 16 + wid] + cr[q * kRouteThreads + t];
    }
    for (int ref = lo; ref < hi; ++ref) {
        int owner, dest;
        keys(ref, owner, dest);
        if (owner == rank) send_ref[ps[dest]++] = ref;
        if (dest == rank) recv_ref[pr[owner]++] = ref;
    }
    for (int k = t; k < W * W; k += kRouteThreads) counts[k] = cnt[k];
}

struct ShardWs { PairWs pair; int32_t *iota; size_t bytes; };
static ShardWs carve_shard_ws(void *base, int B, int d) {
    ShardWs w;
    w.pair = carve_pair_ws(base, B, d, true);
    w.iota = base ? reinterpret_cast<int32_t *>(static_cast<char *>(base) + w.pair.bytes) : nullptr;
    w.bytes = w.pair.bytes + align_up(3 * (size_t)B * 4, 256);
    return w;
}
static inline int grid_for(long long n) { const long long g = (n + 255) / 256; return (int)(g < 4096 ? (g > 0 ? g : 1) : 4096); }
}  // namespace macr

#define MACR_SHARD_COMMON(who)                                                                                         \
    MACR_REQUIRE(B > 0 && dim_supported(d), B > 0 ? MACR_E_UNSUPPORTED : MACR_E_INVALID, who ": B=%d d=%d", B, d);      \
    MACR_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID, who ": workspace"); \
    ShardWs sw = carve_shard_ws(workspace, B, d);                                                                      \
    MACR_REQUIRE(workspace_bytes >= sw.bytes, MACR_E_WORKSPACE, who ": workspace %zu < %zu bytes", workspace_bytes, sw.bytes); \
    const PairWs &ws = sw.pair;                                                                                        \
    hipStream_t st = as_stream(stream)

extern "C" size_t macr_shard_workspace_bytes(int B, int d) {
    if (B <= 0 || !dim_supported(d)) return 0;
    return carve_shard_ws(nullptr, B, d).bytes;
}

namespace macr {
static int validate_lazy(const macr_lazy_adam *lz, bool need_p, bool need_q, const char *who) {
    MACR_REQUIRE(lz->state && (lz->stampP || !need_p) && (lz->stampQ || !need_q), MACR_E_INVALID, "%s: lazy: null pointer", who);
    MACR_REQUIRE(lz->period >= 1 && lz->period <= MACR_LAZY_MAX_PERIOD, MACR_E_INVALID, "%s: lazy period %d outside [1, %d]", who,
                 lz->period, MACR_LAZY_MAX_PERIOD);
    return MACR_OK;
}
static int shard_gather(int B, int d, const float *P_loc, const float *mP, const float *vP, int u_lo, int u_stride, int n_users_loc,
                        const float *Q_loc, const float *mQ, const float *vQ, int i_lo, int i_stride, int n_items_loc,
                        const int32_t *u, const int32_t *i, const int32_t *j, const macr_hyper *hp, const macr_lazy_adam *lz,
                        float *rows3, void *stream) {
    MACR_REQUIRE(B > 0 && dim_supported(d) && P_loc && Q_loc && u && i && j && rows3 && u_lo >= 0 && i_lo >= 0 && u_stride >= 1 &&
                     i_stride >= 1 && n_users_loc >= 0 && n_items_loc >= 0, MACR_E_INVALID, "shard_gather: bad argument");
    const Owned ou = {u_lo, u_stride, n_users_loc}, oi = {i_lo, i_stride, n_items_loc};
    LazyTable tp = {P_loc, mP, vP, nullptr}, tq = {Q_loc, mQ, vQ, nullptr};
    const int grid = grid_for(3LL * B * (d / 4));
    if (lz) {
        MACR_REQUIRE(mP && vP && mQ && vQ && hp, MACR_E_INVALID, "shard_gather_lazy: null pointer");
        if (int e = validate_lazy(lz, n_users_loc > 0, n_items_loc > 0, "shard_gather_lazy")) return e;
        tp.stamp = lz->stampP; tq.stamp = lz->stampQ;
        MACR_DISPATCH_LPR(d, (k_rows_gather_owned<LPR, true><<<grid, 256, 0, as_stream(stream)>>>(
                                 B, tp, ou, tq, oi, u, i, j, rows3, static_cast<const LazyState *>(lz->state), hp->beta1, hp->beta2,
                                 hp->adam_eps)));
    } else {
        MACR_DISPATCH_LPR(d, (k_rows_gather_owned<LPR, false><<<grid, 256, 0, as_stream(stream)>>>(B, tp, ou, tq, oi, u, i, j, rows3,
                                                                                                    nullptr, 0.f, 0.f, 0.f)));
    }
    MACR_CHECK_LAUNCH("shard_gather", as_stream(stream));
    return MACR_OK;
}
}  // namespace macr

extern "C" int macr_shard_gather(int B, int d, const float *P_loc, int u_lo, int u_stride, int n_users_loc, const float *Q_loc,
                                 int i_lo, int i_stride, int n_items_loc, const int32_t *u, const int32_t *i, const int32_t *j,
                                 float *rows3, void *stream) {
    return shard_gather(B, d, P_loc, nullptr, nullptr, u_lo, u_stride, n_users_loc, Q_loc, nullptr, nullptr, i_lo, i_stride,
                        n_items_loc, u, i, j, nullptr, nullptr, rows3, stream);
}

extern "C" int macr_shard_gather_lazy(int B, int d, const float *P_loc, const float *mP, const float *vP, int u_lo, int u_stride,
                                      int n_users_loc, const float *Q_loc, const float *mQ, const float *vQ, int i_lo, int i_stride,
                                      int n_items_loc, const int32_t *u, const int32_t *i, const int32_t *j, const macr_hyper *hp,
                                      const macr_lazy_adam *lazy, float *rows3, void *stream) {
    MACR_REQUIRE(lazy, MACR_E_INVALID, "shard_gather_lazy: lazy is null");
    return shard_gather(B, d, P_loc, mP, vP, u_lo, u_stride, n_users_loc, Q_loc, mQ, vQ, i_lo, i_stride, n_items_loc, u, i, j, hp,
                        lazy, rows3, stream);
}

extern "C" int macr_lazy_rows(long long n, int d, const int32_t *rows, const float *theta, const float *m, const float *v,
                              const uint32_t *stamp, const void *state, const macr_hyper *hp, float *out, void *stream) {
    MACR_REQUIRE(n >= 0 && dim_supported(d), n >= 0 ? MACR_E_UNSUPPORTED : MACR_E_INVALID, "lazy_rows: n=%lld d=%d", n, d);
    MACR_REQUIRE(n == 0 || (rows && theta && m && v && stamp && state && hp && out), MACR_E_INVALID, "lazy_rows: null pointer");
    if (n == 0) return MACR_OK;
    const LazyTable tb = {theta, m, v, stamp};
    const long long groups_per_block = 256 / (d / 4);
    const int grid = (int)((n + groups_per_block - 1) / groups_per_block < 4096 ? (n + groups_per_block - 1) / groups_per_block : 4096);
    MACR_DISPATCH_LPR(d, (k_lazy_rows<LPR><<<grid, 256, 0, as_stream(stream)>>>(n, rows, tb, static_cast<const LazyState *>(state),
                                                                               hp->beta1, hp->beta2, hp->adam_eps, out)));
    MACR_CHECK_LAUNCH("lazy_rows", as_stream(stream));
    return MACR_OK;
}

extern "C" int macr_shard_forward(int loss_kind, int B, int d, const float *rows3, const float *w, const float *wu,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_RUBIBCEBOTH || loss_kind == MACR_LOSS_RUBIBCE || loss_kind == MACR_LOSS_NORMALBCE,
                 MACR_E_INVALID, "shard_forward: loss_kind=%d", loss_kind);
    MACR_REQUIRE(rows3 && w && wu, MACR_E_INVALID, "shard_forward: null pointer");
    MACR_SHARD_COMMON("shard_forward");
    k_iota3<<<grid_for(B), 256, 0, st>>>(B, sw.iota);
    if (loss_kind == MACR_LOSS_NORMALBCE) { MACR_CHECK_LAUNCH("iota", st); return MACR_OK; }   // (forward and backward are one kernel: macr_shard_backward)
    PendingAdam none = {};
    const int user_branch = loss_kind == MACR_LOSS_RUBIBCEBOTH;
    const float *Isrc = rows3 + (size_t)B * d;
    MACR_DISPATCH_LPR(d, (k_pair_fwd<LPR, 0><<<ws.nblk_pair, 256, 0, st>>>(B, ws.Bp, sw.iota, sw.iota + B, sw.iota + 2 * (size_t)B,
                                                                              rows3, Isrc, w, wu, ws.fwd, ws.part, 1, ws.gw, none,
                                                                              user_branch)));
    MACR_CHECK_LAUNCH("pair_fwd", st);
    return MACR_OK;
}

/* rank's share of the (B,B) term; the partial arrays (returned region) must then be summed over the ranks */
extern "C" int macr_shard_bxb(int B, int d, int rank, int world, void **partials, size_t *partial_bytes,
                              void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(world >= 1 && rank >= 0 && rank < world, MACR_E_INVALID, "shard_bxb: rank %d of %d", rank, world);
    MACR_SHARD_COMMON("shard_bxb");
    char *lo = reinterpret_cast<char *>(ws.lpart), *hi = reinterpret_cast<char *>(ws.colpart) + (size_t)ws.nrb * 2 * ws.Bp * 4;
    fill_words(lo, (size_t)(hi - lo) / 4, 0u, st);
    if (partials) *partials = lo;
    if (partial_bytes) *partial_bytes = (size_t)(hi - lo);
    const int rb0 = (int)((long long)ws.nrb * rank / world), rb1 = (int)((long long)ws.nrb * (rank + 1) / world);
    if (rb1 > rb0) {
        BatchSort sort = batch_sort_args(ws, 0, nullptr, nullptr, nullptr);
        sort.rb0 = rb0;
        const int nb = (rb1 - rb0) * ws.ncb;
        const bool full = B % 256 == 0;
        AdamArgs none; none.n_seg = 0;
#define MACR_BXB_ROWS_LAUNCH(R)                                                                                               \
        if (full) k_bxb<R, true, 0><<<nb, 256, 0, st>>>(B, ws.Bp, ws.ncb, nb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, none, ws.scal, sort); \
        else      k_bxb<R, false, 0><<<nb, 256, 0, st>>>(B, ws.Bp, ws.ncb, nb, ws.fwd, ws.rowpart, ws.colpart, ws.lpart, none, ws.scal, sort)
        switch (ws.rows) { case 1: MACR_BXB_ROWS_LAUNCH(1); break; case 2: MACR_BXB_ROWS_LAUNCH(2); break; default: MACR_BXB_ROWS_LAUNCH(4); break; }
#undef MACR_BXB_ROWS_LAUNCH
    }
    MACR_CHECK_LAUNCH("bxb", st);
    return MACR_OK;
}

/* gradient rows of the whole batch into the staging buffer of the workspace; losses (dev) fp32[3]; the branch-vector
 * partial rows (returned region) are replicated work: broadcast rank 0's so that w, w_user stay bit-identical */
extern "C" int macr_shard_backward(int loss_kind, int B, int d, const float *rows3, const float *w, const float *wu,
                                   float *adam_pow, const macr_hyper *hp, float *losses, void **branch_grads,
                                   size_t *branch_bytes, void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_RUBIBCEBOTH || loss_kind == MACR_LOSS_RUBIBCE || loss_kind == MACR_LOSS_NORMALBCE,
                 MACR_E_INVALID, "shard_backward: loss_kind=%d", loss_kind);
    MACR_REQUIRE(rows3 && w && wu && adam_pow && losses, MACR_E_INVALID, "shard_backward: null pointer");
    if (int e = validate_hyper(hp, "shard_backward")) return e;
    MACR_SHARD_COMMON("shard_backward");
    LossArgs L;
    L.part = ws.part; L.n_part = ws.nblk_pair; L.part2 = nullptr; L.n_part2 = 0;
    L.lpart = ws.lpart; L.n_lpart = ws.nrb * ws.ncbx;
    L.kind = loss_kind; L.B = B; L.batch_size_cfg = hp->batch_size_cfg;
    L.alpha = hp->alpha; L.beta = hp->beta; L.decay = hp->decay; L.losses = losses;
    const float coef = hp->decay / (float)hp->batch_size_cfg;
    const float *Isrc = rows3 + (size_t)B * d;
    if (loss_kind == MACR_LOSS_NORMALBCE) {
        // macr_mf/model.py:277-287: no (B,B) term, no branch vectors -- per-pair forward and backward in one kernel, gradient
        // rows of the whole batch into the staging buffer (replicated work, a few MB), then the loss sums
        MACR_DISPATCH_LPR(d, (k_pair_normal_stage<LPR><<<ws.nblk_bwd, 256, 0, st>>>(
                                 B, sw.iota, sw.iota + B, sw.iota + 2 * (size_t)B, rows3, Isrc, ws.stage, ws.part, coef, 1, adam_pow,
                                 adam_pow, ws.scal, hp->lr, hp->beta1, hp->beta2)));
        MACR_CHECK_LAUNCH("pair_normal", st);
        L.n_part = ws.nblk_bwd; L.lpart = nullptr; L.n_lpart = 0;
        k_finalize_losses<<<1, 64, 0, st>>>(L);
        MACR_CHECK_LAUNCH("finalize_losses", st);
        if (branch_grads) *branch_grads = nullptr;
        if (branch_bytes) *branch_bytes = 0;
        return MACR_OK;
    }
    MACR_DISPATCH_LPR(d, (k_pair_bwd_stage<LPR><<<ws.nblk_bwd + 1, 256, 0, st>>>(
                             B, ws.Bp, ws.nrb, ws.ncb, sw.iota, sw.iota + B, sw.iota + 2 * (size_t)B, rows3, Isrc, w, wu, ws.fwd,
                             ws.rowpart, ws.colpart, ws.stage, ws.gw, hp->alpha, hp->beta, coef, adam_pow, ws.scal, hp->lr,
                             hp->beta1, hp->beta2, L)));
    MACR_CHECK_LAUNCH("pair_bwd", st);
    if (branch_grads) *branch_grads = ws.gw;
    if (branch_bytes) *branch_bytes = (size_t)kBranchSlots * 2 * d * 4;
    return MACR_OK;
}

/* ---- the SPLIT step (round 5): forward and backward of a rank's SLICE of the batch only ------------------------------------
 * Rank r runs the per-pair forward and backward for the positions [t0, t1) whose (B,B) row blocks it evaluates anyway
 * (macr_shard_bxb), on rows it received from their owners (all-to-all #1), and sends the gradient rows back to the owners
 * (all-to-all #2) -- a rank moves 2 * 3B/W rows per step instead of taking part in an all-reduce of 3B.  Host side:
 * macr_amd/sharded_train.py::RowShardedMF.step_split.  Branch losses only (the losses with a (B,B) term). */
// the routing tables of the split step in ONE launch (RowShardedMF.route did it with ~10 torch launches: bucketize, three
// owner lookups, a bincount, two stable argsorts).  bounds_u / bounds_i (dev, int32[world], NULL = interleaved rows: owner =
// row %% world): first row NOT owned by rank q; slice_end (dev, int32[world]): end of rank q's slice of batch positions.
extern "C" int macr_shard_route(int B, int world, int rank, const int32_t *u, const int32_t *i, const int32_t *j,
                                const int32_t *bounds_u, const int32_t *bounds_i, const int32_t *slice_end, int32_t *counts,
                                int32_t *send_ref, int32_t *recv_ref, void *stream) {
    MACR_REQUIRE(B > 0 && world >= 1 && world <= kRouteMaxW && rank >= 0 && rank < world, MACR_E_UNSUPPORTED,
                 "shard_route: B=%d world=%d rank=%d (world <= %d)", B, world, rank, kRouteMaxW);
    MACR_REQUIRE(3LL * B <= (1LL << 24), MACR_E_UNSUPPORTED, "shard_route: B=%d", B);
    MACR_REQUIRE(u && i && j && slice_end && counts && send_ref && recv_ref, MACR_E_INVALID, "shard_route: null pointer");
    const RouteOwner ou = {world, bounds_u}, oi = {world, bounds_i};
    const size_t smem = ((size_t)2 * world * kRouteThreads + (size_t)world * world + 2 * world + 2 * world * 16) * 4;
    auto kern = k_shard_route;
    MACR_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess,
                 MACR_E_LAUNCH, "shard_route: cannot reserve %zu B of LDS", smem);
    hipStream_t st = as_stream(stream);
    kern<<<1, kRouteThreads, smem, st>>>(B, world, rank, u, i, j, ou, oi, slice_end, counts, send_ref, recv_ref);
    MACR_CHECK_LAUNCH("shard_route", st);
    return MACR_OK;
}

extern "C" int macr_shard_slice(int B, int d, int rank, int world, int *t0, int *t1) {
    MACR_REQUIRE(B > 0 && dim_supported(d) && world >= 1 && rank >= 0 && rank < world && t0 && t1, MACR_E_INVALID, "shard_slice: bad argument");
    const PairWs ws = carve_pair_ws(nullptr, B, d, true);
    const int rows = 64 * ws.rows;
    const long long a = (long long)ws.nrb * rank / world * rows, b = (long long)ws.nrb * (rank + 1) / world * rows;
    *t0 = (int)(a < B ? a : B);
    *t1 = (int)(b < B ? b : B);
    return MACR_OK;
}

/* forward of the slice: rows3_slice (dev) fp32[3][n][d] = the user, positive and negative rows of positions t0 .. t0+n-1.
 * Writes the slice of the forward arrays and the slice's loss partials into a ZEROED region of the workspace and returns that
 * region: its sum over the ranks is the forward state of the whole batch on every rank (x + 0 = x: exact). */
extern "C" int macr_shard_forward_slice(int loss_kind, int B, int d, int t0, int n, const float *rows3_slice, const float *w,
                                        const float *wu, void **region, size_t *region_bytes, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_RUBIBCEBOTH || loss_kind == MACR_LOSS_RUBIBCE, MACR_E_UNSUPPORTED,
                 "shard_forward_slice: loss_kind=%d (the split step serves the losses with a (B,B) term)", loss_kind);
    MACR_REQUIRE(w && wu && t0 >= 0 && n >= 0 && t0 + n <= B && (n == 0 || rows3_slice), MACR_E_INVALID, "shard_forward_slice: bad argument");
    MACR_SHARD_COMMON("shard_forward_slice");
    char *lo = reinterpret_cast<char *>(ws.fwd), *hi = reinterpret_cast<char *>(ws.lpart);      // fwd | part | part2
    fill_words(lo, (size_t)(hi - lo) / 4, 0u, st);
    fill_words(ws.gw, (size_t)kBranchSlots * 2 * d, 0u, st);
    if (region) *region = lo;
    if (region_bytes) *region_bytes = (size_t)(hi - lo);
    if (n > 0) {
        k_iota3<<<grid_for(n), 256, 0, st>>>(n, sw.iota);
        PendingAdam none = {};
        const int lpr = d / 4, rpb = 256 / lpr, nblk = (n + rpb - 1) / rpb;
        MACR_DISPATCH_LPR(d, (k_pair_fwd<LPR, 0><<<nblk, 256, 0, st>>>(n, ws.Bp, sw.iota, sw.iota + n, sw.iota + 2 * (size_t)n, rows3_slice,
                                                                          rows3_slice + (size_t)n * d, w, wu, ws.fwd + t0, ws.part, 1, ws.gw,
                                                                          none, loss_kind == MACR_LOSS_RUBIBCEBOTH ? 1 : 0)));
    }
    MACR_CHECK_LAUNCH("pair_fwd", st);
    return MACR_OK;
}

/* backward of the slice (after the partial sums of macr_shard_bxb have been summed over the ranks): gradient rows of positions
 * t0 .. t0+n-1 into stage_slice (dev) fp32[3][n][d]; losses of the WHOLE batch (identical on every rank); the branch-vector
 * partial rows (returned region) hold this slice's share: sum them over the ranks. */
extern "C" int macr_shard_backward_slice(int loss_kind, int B, int d, int t0, int n, const float *rows3_slice, const float *w,
                                         const float *wu, float *adam_pow, const macr_hyper *hp, float *losses,
                                         float *stage_slice, void **branch_grads, size_t *branch_bytes, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_RUBIBCEBOTH || loss_kind == MACR_LOSS_RUBIBCE, MACR_E_UNSUPPORTED,
                 "shard_backward_slice: loss_kind=%d", loss_kind);
    MACR_REQUIRE(w && wu && adam_pow && losses && t0 >= 0 && n >= 0 && t0 + n <= B && (n == 0 || (rows3_slice && stage_slice)),
                 MACR_E_INVALID, "shard_backward_slice: bad argument");
    if (int e = validate_hyper(hp, "shard_backward_slice")) return e;
    MACR_SHARD_COMMON("shard_backward_slice");
    LossArgs L;
    L.part = ws.part; L.n_part = ws.nblk_pair; L.part2 = nullptr; L.n_part2 = 0;      // (every rank's blocks were summed index by index)
    L.lpart = ws.lpart; L.n_lpart = ws.nrb * ws.ncbx;
    L.kind = loss_kind; L.B = B; L.batch_size_cfg = hp->batch_size_cfg;
    L.alpha = hp->alpha; L.beta = hp->beta; L.decay = hp->decay; L.losses = losses;
    const float coef = hp->decay / (float)hp->batch_size_cfg;
    const int lpr = d / 4, rpb = 256 / lpr;
    int nblk = (n + rpb - 1) / rpb;
    nblk = nblk < 4096 ? nblk : 4096;
    // (the branch-vector partial rows were zeroed by the forward of the slice; a slice without triples still runs the last block:
    // the step's lr_t, the beta powers and the losses)
    MACR_DISPATCH_LPR(d, (k_pair_bwd_stage<LPR><<<nblk + 1, 256, 0, st>>>(
                             n, ws.Bp, ws.nrb, ws.ncb, sw.iota, sw.iota + n, sw.iota + 2 * (size_t)n, rows3_slice,
                             rows3_slice + (size_t)n * d, w, wu, ws.fwd + t0, ws.rowpart + t0, ws.colpart + t0, stage_slice, ws.gw,
                             hp->alpha, hp->beta, coef, adam_pow, ws.scal, hp->lr, hp->beta1, hp->beta2, L, nullptr, B)));
    MACR_CHECK_LAUNCH("pair_bwd", st);
    if (branch_grads) *branch_grads = ws.gw;
    if (branch_bytes) *branch_bytes = (size_t)kBranchSlots * 2 * d * 4;
    return MACR_OK;
}

/* where macr_shard_apply reads the batch's gradient rows: (dev) fp32[3][B][d], row role * B + t = the gradient of position t's
 * user / positive / negative row.  The split step fills the rows this rank owns from what all-to-all #2 delivered. */
extern "C" int macr_shard_stage(int B, int d, float **stage, void *workspace, size_t workspace_bytes) {
    MACR_REQUIRE(stage, MACR_E_INVALID, "shard_stage: null pointer");
    void *stream = nullptr;
    MACR_SHARD_COMMON("shard_stage");
    (void)st;
    *stage = ws.stage;
    return MACR_OK;
}

/* this rank's rows: sort the references to them, one owner per row sums its staging rows, dense Adam over the shard
 * (lazy != NULL: the lazy pass -- the batch's rows and this step's K-th of the shard) */
namespace macr {
static int shard_apply(int loss_kind, int B, int d, int n_users_loc, int n_items_loc, int u_lo, int u_stride, int i_lo,
                       int i_stride, const int32_t *u, const int32_t *i, const int32_t *j, float *P, float *Q, float *w,
                       float *wu, float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu, float *vwu,
                       float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ, const macr_hyper *hp,
                       const macr_lazy_adam *lz, void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(n_users_loc >= 0 && n_items_loc >= 0 && u_stride >= 1 && i_stride >= 1 && u && i && j && P && Q && w && wu && mP &&
                     vP && mQ && vQ && mw && vw && mwu && vwu && gP && gQ && touchedP && touchedQ, MACR_E_INVALID,
                 "shard_apply: bad argument");
    if (int e = validate_hyper(hp, "shard_apply")) return e;
    if (lz) if (int e = validate_lazy(lz, n_users_loc > 0, n_items_loc > 0, "shard_apply_lazy")) return e;
    MACR_SHARD_COMMON("shard_apply");
    const int n = 3 * B;
    const Owned ou = {u_lo, u_stride, n_users_loc}, oi = {i_lo, i_stride, n_items_loc};
    k_shard_keys<<<grid_for(B), 256, 0, st>>>(B, ou, oi, u, i, j, ws.ska, ws.sva, ws.n_work,
                                              lz ? static_cast<LazyState *>(lz->state) : nullptr, ws.scal);
    const int flip = launch_radix_sort(ws.ska, ws.sva, ws.skb, ws.svb, n, (uint32_t)(n_users_loc + n_items_loc), ws.ghist, st);
    MACR_CHECK_LAUNCH("ref_sort", st);
    const uint32_t *sk = flip ? ws.skb : ws.ska, *sv = flip ? ws.svb : ws.sva;
    // The step completes here, so nothing needs a row's gradient as a row in memory: k_seg_scan finds each row's first
    // reference and names its run in the row's flag, the Adam pass sums the <= 16 staged rows itself (adam_block INDEXED;
    // longer runs -- the hot items -- go through gP/gQ by k_seg_sum).  MACR_SEG_UNFUSED=1: the segment reduce writes
    // every row, as before (A/B measurements, tests).
    const char *unfused = getenv("MACR_SEG_UNFUSED");
    const bool indexed = (size_t)n <= kRefMaxRefs && !(unfused && unfused[0] == '1');
    if (indexed) {
        uint32_t *work = flip ? ws.ska : ws.skb;                       // the buffer the sort no longer needs
        k_seg_scan<<<(n + 255) / 256, 256, 0, st>>>(n, n_users_loc, (uint32_t)(n_users_loc + n_items_loc), sk, touchedP, touchedQ, work, ws.n_work);
        MACR_CHECK_LAUNCH("seg_index", st);
        MACR_DISPATCH_LPR(d, (k_seg_sum<LPR><<<1024, 256, 0, st>>>(work, ws.n_work, n_users_loc, sk, sv, ws.stage, gP, gQ)));
        MACR_CHECK_LAUNCH("seg_sum", st);
    } else {
        MACR_DISPATCH_LPR(d, (k_seg_reduce<LPR><<<(n + 256 / LPR - 1) / (256 / LPR), 256, 0, st>>>(
                                 n, n_users_loc, (uint32_t)(n_users_loc + n_items_loc), sk, sv, ws.stage, gP, gQ, touchedP, touchedQ)));
        MACR_CHECK_LAUNCH("seg_reduce", st);
    }
    AdamArgs a;
    long long nb = 0;
    a.n_seg = 0; a.lpr = d / 4; a.lpr_shift = d == 32 ? 3 : d == 64 ? 4 : d == 128 ? 5 : 6;
    a.b1 = hp->beta1; a.b2 = hp->beta2; a.eps = hp->adam_eps;
    if (n_users_loc) add_seg(a, P, mP, vP, gP, touchedP, n_users_loc, nb);
    if (n_items_loc) add_seg(a, Q, mQ, vQ, gQ, touchedQ, n_items_loc, nb);
    const int n_tab = a.n_seg;
    if (loss_kind != MACR_LOSS_NORMALBCE) add_seg(a, w, mw, vw, ws.gw, nullptr, 1, nb, kBranchSlots, 2 * d);
    if (loss_kind == MACR_LOSS_RUBIBCEBOTH) add_seg(a, wu, mwu, vwu, ws.gw + d, nullptr, 1, nb, kBranchSlots, 2 * d);
    LossArgs L; L.losses = nullptr;
    if (indexed)
        for (int k = 0; k < n_tab; ++k) { a.seg[k].sv = sv; a.seg[k].stage = ws.stage; }
    if (lz) {
        // (a rank without user rows has its item rows in segment 0: the list's keys are [0, n_users_loc) | [n_users_loc, end) either way)
        LazyArgs z;
        z.state = static_cast<const LazyState *>(lz->state); z.period = lz->period; z.n_tab = n_tab;
        long long at = 0;
        for (int k = 0; k < n_tab; ++k) {
            a.seg[k].stamp = a.seg[k].theta == P ? lz->stampP : lz->stampQ;
            const long long chunks = (a.seg[k].n_vec + kAdamVecPerBlock - 1) / kAdamVecPerBlock;
            z.sweep_first[k] = at;
            at += (chunks + lz->period - 1) / lz->period;
        }
        for (int k = n_tab; k < 3; ++k) z.sweep_first[k] = at;
        z.touch_first = at;
        at += (n + (256 / a.lpr) - 1) / (256 / a.lpr);
        z.branch_first = at;
        at += a.n_seg - n_tab;
        z.n_refs = n; z.n_users = n_users_loc; z.item_seg = n_users_loc ? 1 : 0;
        z.key_end = (uint32_t)(n_users_loc + n_items_loc); z.sk = sk;
        if (indexed) k_adam_lazy<true, true><<<(unsigned)at, 256, 0, st>>>(a, z, ws.scal);
        else         k_adam_lazy<false, true><<<(unsigned)at, 256, 0, st>>>(a, z, ws.scal);
        MACR_CHECK_LAUNCH("adam_lazy", st);
        return MACR_OK;
    }
    if (indexed) {
        k_adam_dense<true><<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
        MACR_CHECK_LAUNCH("adam_indexed", st);
        return MACR_OK;
    }
    k_adam_dense<false><<<(unsigned)nb, 256, 0, st>>>(a, ws.scal, L);
    MACR_CHECK_LAUNCH("adam_dense", st);
    return MACR_OK;
}
}  // namespace macr

extern "C" int macr_shard_apply(int loss_kind, int B, int d, int n_users_loc, int n_items_loc, int u_lo, int u_stride, int i_lo,
                                int i_stride, const int32_t *u, const int32_t *i, const int32_t *j, float *P, float *Q, float *w,
                                float *wu, float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu, float *vwu,
                                float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ, const macr_hyper *hp,
                                void *workspace, size_t workspace_bytes, void *stream) {
    return shard_apply(loss_kind, B, d, n_users_loc, n_items_loc, u_lo, u_stride, i_lo, i_stride, u, i, j, P, Q, w, wu, mP, vP, mQ, vQ,
                       mw, vw, mwu, vwu, gP, gQ, touchedP, touchedQ, hp, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int macr_shard_apply_lazy(int loss_kind, int B, int d, int n_users_loc, int n_items_loc, int u_lo, int u_stride, int i_lo,
                                     int i_stride, const int32_t *u, const int32_t *i, const int32_t *j, float *P, float *Q, float *w,
                                     float *wu, float *mP, float *vP, float *mQ, float *vQ, float *mw, float *vw, float *mwu,
                                     float *vwu, float *gP, float *gQ, int32_t *touchedP, int32_t *touchedQ, const macr_hyper *hp,
                                     const macr_lazy_adam *lazy, void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(lazy, MACR_E_INVALID, "shard_apply_lazy: lazy is null");
    return shard_apply(loss_kind, B, d, n_users_loc, n_items_loc, u_lo, u_stride, i_lo, i_stride, u, i, j, P, Q, w, wu, mP, vP, mQ, vQ,
                       mw, vw, mwu, vwu, gP, gQ, touchedP, touchedQ, hp, lazy, workspace, workspace_bytes, stream);
}

/* every row of the (local) tables brought to the current step: afterwards P, Q and the slots are what the per-step dense pass
 * leaves (before evaluation, checkpoints, anything that reads the tables) */
extern "C" int macr_lazy_flush(int d, long long n_rows_p, long long n_rows_q, float *P, float *Q, float *mP, float *vP, float *mQ,
                               float *vQ, const macr_hyper *hp, const macr_lazy_adam *lazy, void *stream) {
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "lazy_flush: d=%d", d);
    MACR_REQUIRE(n_rows_p >= 0 && n_rows_q >= 0 && lazy && (n_rows_p == 0 || (P && mP && vP)) && (n_rows_q == 0 || (Q && mQ && vQ)),
                 MACR_E_INVALID, "lazy_flush: bad argument");
    if (int e = validate_hyper(hp, "lazy_flush")) return e;
    if (int e = validate_lazy(lazy, n_rows_p > 0, n_rows_q > 0, "lazy_flush")) return e;
    return lazy_flush_tables(d, n_rows_p, n_rows_q, P, Q, mP, vP, mQ, vQ, hp, lazy, as_stream(stream));
}

// ---- LightGCN ---------------------------------------------------------------
namespace macr {
struct PlanHeaderLite { int32_t magic, n_items, n_split, n_slots, N, chunk, n_groups, reserved; };   // = spmm_kernels.hip PlanHeader
struct LgcnWs { float *E, *dE, *G, *work; int32_t *cnt; double *emb_acc; PairWs pair; size_t bytes; };
static LgcnWs carve_lgcn_ws(void *base, int B, int N, int d, const PlanHeaderLite *ph, const PlanHeaderLite *pht = nullptr) {
    LgcnWs w;
    char *p = static_cast<char *>(base);
    const size_t nd = align_up((size_t)N * d * 4, 256);
    size_t off = 0;
    auto take = [&](size_t bytes) { void *r = p ? p + off : nullptr; off += bytes; return r; };
    w.E = static_cast<float *>(take(nd));
    w.dE = static_cast<float *>(take(nd));
    w.G = static_cast<float *>(take(nd));
    // layer buffers, partial rows of the hub pieces, their arrival counters (= macr_lgcn_work_floats; zero between steps)
    // (pht: the plan of the transposed adjacency of an asymmetric --adj_type; the two propagations never run at once and share the buffers)
    const size_t wf = macr_lgcn_work_floats(N, d, ph), wft = pht ? macr_lgcn_work_floats(N, d, pht) : 0;
    w.work = static_cast<float *>(take(align_up((wf > wft ? wf : wft) * 4, 256)));
    w.cnt = static_cast<int32_t *>(take(align_up((size_t)N * 4, 256)));    // references of the current batch per row (zero between steps)
    w.emb_acc = static_cast<double *>(take(kEmbSlots * 8));                 // emb_loss partial sums (zero between steps)
    w.pair = carve_pair_ws(p ? p + off : nullptr, B, d);
    off += w.pair.bytes;
    w.bytes = off;
    return w;
}
}  // namespace macr


extern "C" size_t macr_lgcn_train_workspace_bytes(int B, int N, int d, const void *plan_host) {
    if (B <= 0 || N <= 0 || !dim_supported(d)) return 0;
    const PlanHeaderLite *ph = static_cast<const PlanHeaderLite *>(plan_host);
    return carve_lgcn_ws(nullptr, B, N, d, ph).bytes;
}
extern "C" size_t macr_lgcn_train_workspace_bytes_t(int B, int N, int d, const void *plan_host, const void *plan_t_host) {
    if (B <= 0 || N <= 0 || !dim_supported(d)) return 0;
    return carve_lgcn_ws(nullptr, B, N, d, static_cast<const PlanHeaderLite *>(plan_host), static_cast<const PlanHeaderLite *>(plan_t_host)).bytes;
}

// rowptr_t / col_t / val_t / plan_t_*: the TRANSPOSED adjacency, used by the backward propagation (the gradient of A E is A^T dE:
// tf.gradients of tf.sparse_tensor_dense_matmul, LightGCN.py:301).  For the symmetric `pre` / `plain` matrices it is A itself
// (macr_lgcn_train_step); --adj_type norm / gcmc / mean are D^-1 A (utility/load_data.py:95-164, LightGCN.py:667-678).
extern "C" int macr_lgcn_train_step_t(int loss_kind, int B, int d, int n_users, int n_items, int n_layers,
                                    const int32_t *rowptr, const int32_t *col, const float *val,
                                    const void *plan_dev, const void *plan_host,
                                    const int32_t *rowptr_t, const int32_t *col_t, const float *val_t,
                                    const void *plan_t_dev, const void *plan_t_host, const int32_t *u,
                                    const int32_t *i, const int32_t *j, float *T, float *w, float *wu, float *mT,
                                    float *vT, float *mw, float *vw, float *mwu, float *vwu, float *adam_pow,
                                    const macr_hyper *hp, float *losses, int flags, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(loss_kind == MACR_LOSS_NORMALBCE || loss_kind == MACR_LOSS_RUBIBCEBOTH, MACR_E_INVALID,
                 "lgcn_train_step: loss_kind=%d", loss_kind);
    MACR_REQUIRE(B > 0 && n_users > 0 && n_items > 0 && n_layers >= 0, MACR_E_INVALID,
                 "lgcn_train_step: B=%d n_users=%d n_items=%d n_layers=%d", B, n_users, n_items, n_layers);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "lgcn_train_step: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(rowptr && col && val && rowptr_t && col_t && val_t && u && i && j && T && mT && vT && adam_pow && losses && workspace,
                 MACR_E_INVALID, "lgcn_train_step: null pointer");
    MACR_REQUIRE((plan_t_dev == nullptr) == (plan_t_host == nullptr) && (plan_t_dev == nullptr) == (plan_dev == nullptr), MACR_E_INVALID,
                 "lgcn_train_step: the transposed adjacency comes with a plan when the adjacency does");
    MACR_REQUIRE(w && wu && mw && vw && mwu && vwu, MACR_E_INVALID, "lgcn_train_step: null branch vectors");
    MACR_REQUIRE((flags & ~(MACR_STEP_LOSS_ONLY | MACR_STEP_DENSE_LAYERS)) == 0, MACR_E_INVALID, "lgcn_train_step: flags=%d", flags);
    const bool loss_only = flags & MACR_STEP_LOSS_ONLY;
    if (int e = validate_hyper(hp, "lgcn_train_step")) return e;
    const int N = n_users + n_items;
    MACR_REQUIRE((plan_dev == nullptr) == (plan_host == nullptr), MACR_E_INVALID,
                 "lgcn_train_step: plan needs both its device copy and its host copy (or neither)");
    const PlanHeaderLite *ph = static_cast<const PlanHeaderLite *>(plan_host);
    MACR_REQUIRE(!ph || ph->N == N, MACR_E_INVALID, "lgcn_train_step: plan does not belong to this graph");
    const PlanHeaderLite *pht = static_cast<const PlanHeaderLite *>(plan_t_host);
    MACR_REQUIRE(!pht || pht->N == N, MACR_E_INVALID, "lgcn_train_step: transposed plan does not belong to this graph");
    LgcnWs ws = carve_lgcn_ws(workspace, B, N, d, ph, pht == ph ? nullptr : pht);
    MACR_REQUIRE(workspace_bytes >= ws.bytes, MACR_E_WORKSPACE, "lgcn_train_step: workspace %zu < %zu bytes",
                 workspace_bytes, ws.bytes);
    MACR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID,
                 "lgcn_train_step: workspace must be 256-byte aligned");
    hipStream_t st = as_stream(stream);
    const size_t nd = (size_t)N * d;
    // The step reads the propagated table at the <= 3B rows of its batch only, and its gradient enters the backward
    // propagation with the same <= 3B non-zero rows: the last forward layer is computed for those rows only and the first
    // backward layer gathers only from them (spmm_kernels.hip kSparseOut / kSparseIn).  MACR_STEP_DENSE_LAYERS, or
    // MACR_LGCN_DENSE=1 in the environment, keeps every layer dense (the forward result is bit-identical either way).
    static const bool dense_layers = getenv("MACR_LGCN_DENSE") && getenv("MACR_LGCN_DENSE")[0] == '1';
    // (one layer: it would be the row-sparse first AND the fused last backward layer at once, reading the reference counts
    // as flags while other waves clear them -- such a model runs its layers dense)
    const bool sparse = !dense_layers && !(flags & MACR_STEP_DENSE_LAYERS) && n_layers > 1;
    // The forward's last layer has one wave per batch reference; in a training step those waves also COUNT the references
    // per row (ws.cnt): the flags of the first backward layer and the multiplicities of the ego-row regulariser.
    // Positives repeat (a popularity-skewed batch refers to its hottest item hundreds of times, and that many atomics
    // on one counter queue up): they are counted where the batch is already grouped -- one atomic per distinct row of
    // a 16-slot chunk in pair_bwd / pair_normal -- unless the batch takes the staged path (count = 2: all three here).
    const SparseCtx sp = {ws.cnt, u, i, j, B, n_users, ph ? ph->chunk : 0x7fffffff, loss_only ? 0 : ws.pair.staged ? 2 : 1};
    // forward propagation (LightGCN.py:288-309)
    if (int e = launch_propagate(N, d, n_layers, rowptr, col, val, plan_dev, plan_host, T, ws.E, ws.work, st,
                                 sparse ? &sp : nullptr, kSparseOut, nullptr))
        return e;
    LossArgs L;
    L.part = ws.pair.part;
    L.part2 = ws.pair.part2; L.n_part2 = ws.pair.nblk_bwd;
    L.lpart = ws.pair.lpart; L.n_lpart = loss_kind == MACR_LOSS_RUBIBCEBOTH ? ws.pair.nrb * ws.pair.ncbx : 0;
    L.kind = loss_kind; L.B = B; L.batch_size_cfg = hp->batch_size_cfg;
    L.alpha = hp->alpha; L.beta = hp->beta; L.decay = hp->decay; L.losses = losses;
    const float coef = hp->decay / (float)hp->batch_size_cfg;
    if (loss_only) {
        // the reference's "test loss" pass (LightGCN.py:799-819): loss_X, mf_loss_X, emb_loss_X without opt_X
        float *Ei0 = ws.E + (size_t)n_users * d;
        if (int e = launch_pair(loss_kind, B, d, n_users, n_items, u, i, j, ws.E, Ei0, w, wu, nullptr, nullptr, nullptr,
                                nullptr, 0.0f, 0, adam_pow, hp, ws.pair, st, nullptr, nullptr, 0, nullptr, true))
            return e;
        MACR_DISPATCH_D(d, (k_reg_scatter<D><<<ws.pair.nblk_bwd, 256, 0, st>>>(B, n_users, u, i, j, T, nullptr, coef, ws.pair.part2)));
        MACR_CHECK_LAUNCH("reg_scatter", st);
        L.n_part = loss_kind == MACR_LOSS_NORMALBCE
                       ? ((B + kChunkT - 1) / kChunkT < 1024 ? (B + kChunkT - 1) / kChunkT : 1024) : ws.pair.nblk_pair;
        k_finalize_losses<<<1, 64, 0, st>>>(L);
        MACR_CHECK_LAUNCH("finalize_losses", st);
        return MACR_OK;
    }
    // dE: the gradient w.r.t. the propagated table; the pair kernels ADD into the rows of the batch.  Dense layers: the
    // whole buffer is cleared here (and again on return); sparse layers: every row is zero already (the fused epilogue of the
    // previous step cleared the batch's rows, the workspace starts zeroed).
    if (!sparse) fill_words(ws.dE, nd, 0u, st);
    // pair loss on the propagated rows; items live at rows n_users.. of E
    float *Ei = ws.E + (size_t)n_users * d, *dEi = ws.dE + (size_t)n_users * d;
    if (int e = launch_pair(loss_kind, B, d, n_users, n_items, u, i, j, ws.E, Ei, w, wu, ws.dE, dEi, nullptr, nullptr,
                            0.0f, 0, adam_pow, hp, ws.pair, st, nullptr, nullptr, 0, nullptr, false,
                            sparse && !ws.pair.staged ? ws.cnt + n_users : nullptr))
        return e;
    AdamArgs a;
    a.n_seg = 0;
    a.lpr = d / 4; a.lpr_shift = d == 32 ? 3 : d == 64 ? 4 : d == 128 ? 5 : 6;
    a.b1 = hp->beta1; a.b2 = hp->beta2; a.eps = hp->adam_eps;
    long long nb = 0;
    L.n_part = loss_kind == MACR_LOSS_NORMALBCE ? ws.pair.nblk_bwd : ws.pair.nblk_pair;
    if (sparse) {
        // backward through the propagation with the optimizer in the last layer's epilogue (spmm_kernels.hip AdamFuse):
        // gradient row + ego-row regulariser (LightGCN.py:525-528) -> Adam on T, no G, no separate pass over the table
        const AdamFuse fuse = {T, mT, vT, ws.pair.scal, hp->beta1, hp->beta2, hp->adam_eps, coef, ws.dE, ws.emb_acc};
        SparseCtx spt = sp;
        spt.chunk = pht ? pht->chunk : 0x7fffffff;
        if (int e = launch_propagate(N, d, n_layers, rowptr_t, col_t, val_t, plan_t_dev, plan_t_host, ws.dE, ws.G, ws.work, st, &spt,
                                     kSparseIn, &fuse))
            return e;
        if (loss_kind == MACR_LOSS_RUBIBCEBOTH) {
            add_seg(a, w, mw, vw, ws.pair.gw, nullptr, 1, nb, kBranchSlots, 2 * d);
            add_seg(a, wu, mwu, vwu, ws.pair.gw + d, nullptr, 1, nb, kBranchSlots, 2 * d);
        }
        L.n_part2 = 0;                                  // emb_loss comes from ws.emb_acc
        k_lgcn_finalize<<<a.n_seg + 1, 256, 0, st>>>(a, ws.pair.scal, L, ws.emb_acc);
        MACR_CHECK_LAUNCH("lgcn_finalize", st);
        return MACR_OK;
    }
    // backward through the propagation: the transposed operator (A itself for the symmetric matrices)
    if (int e = launch_propagate(N, d, n_layers, rowptr_t, col_t, val_t, plan_t_dev, plan_t_host, ws.dE, ws.G, ws.work, st, nullptr,
                                 kSparseIn, nullptr))
        return e;
    // l2 regulariser on the ego rows (LightGCN.py:525-528)
    // (the batch grouped by positive item, as the pair launch left it in the workspace, when there is one)
    const int32_t *gu = ws.pair.staged ? u : ws.pair.us, *gi = ws.pair.staged ? i : ws.pair.is, *gj = ws.pair.staged ? j : ws.pair.js;
    MACR_DISPATCH_D(d, (k_reg_scatter<D><<<ws.pair.nblk_bwd, 256, 0, st>>>(B, n_users, gu, gi, gj, T, ws.G, coef, ws.pair.part2)));
    MACR_CHECK_LAUNCH("reg_scatter", st);
    add_seg(a, T, mT, vT, ws.G, nullptr, N, nb);
    if (loss_kind == MACR_LOSS_RUBIBCEBOTH) {
        add_seg(a, w, mw, vw, ws.pair.gw, nullptr, 1, nb, kBranchSlots, 2 * d);
        add_seg(a, wu, mwu, vwu, ws.pair.gw + d, nullptr, 1, nb, kBranchSlots, 2 * d);
    }
    k_adam_dense<false><<<(unsigned)nb, 256, 0, st>>>(a, ws.pair.scal, L);
    MACR_CHECK_LAUNCH("adam_dense", st);
    // (dE is zero again on return, as after a step with the sparse layers: their first backward layer may read all of it)
    fill_words(ws.dE, nd, 0u, st);
    return MACR_OK;
}

// the symmetric case (A^T = A): --adj_type pre (every README command) and plain
extern "C" int macr_lgcn_train_step(int loss_kind, int B, int d, int n_users, int n_items, int n_layers,
                                    const int32_t *rowptr, const int32_t *col, const float *val,
                                    const void *plan_dev, const void *plan_host, const int32_t *u,
                                    const int32_t *i, const int32_t *j, float *T, float *w, float *wu, float *mT,
                                    float *vT, float *mw, float *vw, float *mwu, float *vwu, float *adam_pow,
                                    const macr_hyper *hp, float *losses, int flags, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return macr_lgcn_train_step_t(loss_kind, B, d, n_users, n_items, n_layers, rowptr, col, val, plan_dev, plan_host, rowptr, col, val,
                                  plan_dev, plan_host, u, i, j, T, w, wu, mT, vT, mw, vw, mwu, vwu, adam_pow, hp, losses, flags,
                                  workspace, workspace_bytes, stream);
}
