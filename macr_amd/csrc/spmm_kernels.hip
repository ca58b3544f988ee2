// LightGCN normalised-adjacency SpMM for gfx950.
//
// Replaces the tf.sparse_tensor_dense_matmul loop of
// macr_lightgcn/LightGCN.py:297-305 (100 row folds x n_layers) and the
// stack/reduce_mean of :306-307.  A_hat = D^-1/2 A D^-1/2 in CSR
// (macr_lightgcn/utility/load_data.py:112-121), X is the (N,d) embedding table.
//
// One wave per WORK ITEM; the wave is split into 64/LPR neighbour groups, each
// group streams one neighbour row as LPR lanes x float4 (a coalesced 256-B read at
// d=64), two neighbours in flight per group.  The groups' partial rows are combined
// with cross-lane shuffles; group 0 writes the row.
// Work items come from a static plan (the graph never changes): a row with at most
// kChunk non-zeros is one item and is written with the fused epilogue; a longer row
// (interaction graphs have hub items with 10^4..10^5 neighbours -- Addressa item 0 has
// 7077, the Zipf synthetic Yelp-size graph 1.2e5) is cut into kChunk-sized items whose
// partial rows go to a scratch slab and are summed, in slot order (deterministic, no
// atomics), by a small fix-up kernel that also applies the epilogue.  Without the split
// one wave serialises the whole hub row: 1.65 ms per layer instead of ~0.1 ms.
// HBM-bound: compulsory bytes per layer nnz*8 + (N+1)*4 + 2*N*d*4;
// X itself is Infinity-Cache resident for every real dataset.
#include "common.hpp"

#include <string.h>
#include <vector>

namespace macr {

constexpr int kChunk = 512;       // non-zeros per work item

struct PlanHeader {               // all int32, followed by the arrays below
    int32_t magic, n_items, n_split, n_slots, N, reserved[3];
};
constexpr int32_t kPlanMagic = 0x4d414352;   // "MACR"
// layout after the header:  item_row[n_items] item_beg[n_items] item_end[n_items] item_slot[n_items]
//                           split_row[n_split] split_slot0[n_split+1]

struct PlanView {
    const int32_t *item_row, *item_beg, *item_end, *item_slot, *split_row, *split_slot0;
    int n_items, n_split, n_slots;
};

__device__ __host__ inline PlanView view_plan(const void *plan, const PlanHeader &h) {
    const int32_t *p = reinterpret_cast<const int32_t *>(plan) + sizeof(PlanHeader) / 4;
    PlanView v;
    v.n_items = h.n_items; v.n_split = h.n_split; v.n_slots = h.n_slots;
    v.item_row = p; p += h.n_items;
    v.item_beg = p; p += h.n_items;
    v.item_end = p; p += h.n_items;
    v.item_slot = p; p += h.n_items;
    v.split_row = p; p += h.n_split;
    v.split_slot0 = p;
    return v;
}

// Sum of val[e] * X[col[e]] over e in [beg, end) for one row (or one 512-nnz slice of a hub row), by one wave: the
// 64/LPR lane groups take neighbours grp, grp+NG, ... (each group in ascending order: the summation tree is fixed).
// The wave first fetches the indices and values of up to 64 neighbours with ONE coalesced load each (lane l <- entry
// beg+l) and hands them to the groups by shuffles, then keeps FOUR row gathers in flight per group.  The first
// version loaded col/val per group (16 lanes reading the same word, a dependent trip in front of every pair of row
// gathers) and kept two gathers in flight: a wave with the average 39 neighbours needed ~10 dependent trips.
// Measured on Yelp2018 shapes (tools/bench_lgcn.py): 75 -> 56.5 us per layer, LightGCN step 376 -> 305 us.  Depth 2: 59 us,
// 8: 67 us, 16: 96 us (padding entries of the last round gather too); guarding the padded gathers with a wave-uniform
// test costs more than it saves (70 us: the guards become branches with their own waits).
#ifdef MACR_ABL_SPMM_V1
template <int LPR>
__device__ __forceinline__ float4 gather_range(const int32_t *__restrict__ col, const float *__restrict__ val,
                                               const float *__restrict__ X, int beg, int end, int sub, int grp) {
    constexpr int d = 4 * LPR;
    constexpr int NG = kWave / LPR;                 // neighbour groups per wave
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = beg + grp;
    for (; e + NG < end; e += 2 * NG) {             // two neighbours per group in flight
        const int c0 = col[e], c1 = col[e + NG];
        const float a0 = val[e], a1 = val[e + NG];
        const float4 x0 = ld4(X + (size_t)c0 * d + 4 * sub);
        const float4 x1 = ld4(X + (size_t)c1 * d + 4 * sub);
        acc = fma4(a0, x0, acc);
        acc = fma4(a1, x1, acc);
    }
    if (e < end) acc = fma4(val[e], ld4(X + (size_t)col[e] * d + 4 * sub), acc);
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
    }
    return acc;
}
#else
template <int LPR>
__device__ __forceinline__ float4 gather_range(const int32_t *__restrict__ col, const float *__restrict__ val,
                                               const float *__restrict__ X, int beg, int end, int sub, int grp) {
    constexpr int d = 4 * LPR;
    constexpr int NG = kWave / LPR;                 // neighbour groups per wave
    constexpr int PER = kWave / NG;                 // neighbours per group in a 64-entry chunk (= LPR)
#ifndef MACR_SPMM_DEPTH
#define MACR_SPMM_DEPTH 4
#endif
    constexpr int DEPTH = MACR_SPMM_DEPTH < PER ? MACR_SPMM_DEPTH : PER;      // row gathers in flight per group
    const int lane = threadIdx.x & 63;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += kWave) {           // wave-uniform
        const int n = end - base < kWave ? end - base : kWave;  // entries of this chunk
        const int ec = base + (lane < n ? lane : n - 1);
        const int cv = col[ec];
        const float av = lane < n ? val[ec] : 0.f;              // padding entries add 0 * X[valid row]
        // group grp takes entries grp, grp+NG, ... of the chunk; rounds of DEPTH gathers in flight
#pragma unroll
        for (int k0 = 0; k0 < PER; k0 += DEPTH) {
            if (k0 * NG >= n) break;                            // wave-uniform: nothing left in this chunk
            int c[DEPTH]; float a[DEPTH]; float4 x[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int src = (k0 + k) * NG + grp;            // < 64: PER is a multiple of DEPTH
                c[k] = __shfl(cv, src, kWave); a[k] = __shfl(av, src, kWave);
            }
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) x[k] = ld4(X + (size_t)c[k] * d + 4 * sub);
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) acc = fma4(a[k], x[k], acc);
        }
    }
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
    }
    return acc;
}
#endif

// gather_range for a ROW-SPARSE X: only rows with active[row] != 0 are non-zero (the gradient of a LightGCN batch
// touches <= 3B rows of the table), so the other neighbours are dropped before anything is gathered.  Per 64-entry
// chunk: one byte gather of the flags, a ballot, the surviving (col, val) pairs are packed into the wave's 512 bytes
// of LDS in entry order and the groups take them from there.  A chunk without an active neighbour costs its index
// load and the flag gather.  (The summation tree differs from gather_range's -- same terms, other grouping.)
template <int LPR>
__device__ __forceinline__ float4 gather_range_active(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                      const float *__restrict__ X, const uint8_t *__restrict__ active,
                                                      int beg, int end, int sub, int grp, int2 *s_ent) {
    constexpr int d = 4 * LPR;
    constexpr int NG = kWave / LPR;
    constexpr int PER = kWave / NG;
    constexpr int DEPTH = 4 < PER ? 4 : PER;
    const int lane = threadIdx.x & 63;
    const uint64_t below = (1ull << lane) - 1ull;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += kWave) {           // wave-uniform
        const int n = end - base < kWave ? end - base : kWave;
        const int ec = base + (lane < n ? lane : n - 1);
        const int cv = col[ec];
        const bool act = lane < n && active[cv] != 0;
        const uint64_t m = __ballot(act);
        if (m == 0ull) continue;                                // wave-uniform
        const float av = val[ec];
        const int na = __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous chunk's reads of s_ent are done
        if (act) s_ent[__popcll(m & below)] = make_int2(cv, __float_as_int(av));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int2 pad = s_ent[0];                              // a valid row for the unguarded gathers of padding slots
#pragma unroll
        for (int k0 = 0; k0 < PER; k0 += DEPTH) {
            if (k0 * NG >= na) break;                           // wave-uniform
            int c[DEPTH]; float a[DEPTH]; float4 x[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const int src = (k0 + k) * NG + grp;
                const int2 e = s_ent[src < na ? src : 0];
                c[k] = src < na ? e.x : pad.x;
                a[k] = src < na ? __int_as_float(e.y) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) x[k] = ld4(X + (size_t)c[k] * d + 4 * sub);
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) acc = fma4(a[k], x[k], acc);
        }
    }
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
    }
    return acc;
}

// s_on: 0 = S_in[r] is to be taken as zero (a row-sparse S_in whose inactive rows were never written)
template <int LPR>
__device__ __forceinline__ void write_row(int r, int sub, float4 acc, float *Y, const float *S_in, float *S_out, float scale,
                                          bool s_on = true) {
    constexpr int d = 4 * LPR;
    const size_t o = (size_t)r * d + 4 * sub;
    if (Y) st4(Y + o, acc);
    if (S_out) {
        const float4 s = s_on ? ld4(S_in + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        st4(S_out + o, make_float4((s.x + acc.x) * scale, (s.y + acc.y) * scale, (s.z + acc.z) * scale, (s.w + acc.w) * scale));
    }
}

// Y = A X (if Y), S_out = (S_in + A X) * scale (if S_out; may alias S_in: updated in place row by row).
// SPARSE (a LightGCN training step only needs the propagated rows of its batch, and its gradient enters the backward
// propagation with <= 3B non-zero rows; rows[] = 1 for those rows):
//   kSparseOut  only rows with rows[r] != 0 are computed (the LAST forward layer: nothing else is read afterwards)
//   kSparseIn   X and S_in are row-sparse: only rows with rows[.] != 0 are read (the FIRST backward layer)
constexpr int kDense = 0, kSparseOut = 1, kSparseIn = 2;
template <int LPR, int SPARSE>
__global__ __launch_bounds__(256) void k_spmm_csr(int N, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const float *__restrict__ val,
                                                  const void *__restrict__ plan, PlanHeader ph,
                                                  const float *__restrict__ X, float *Y, const float *S_in,
                                                  float *S_out, float scale, float *__restrict__ slab,
                                                  const uint8_t *__restrict__ rows) {
    constexpr int d = 4 * LPR;
    __shared__ int2 s_ent[SPARSE == kSparseIn ? 4 * kWave : 1];
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    int r, beg, end, slot = -1;
    if (plan) {
        const PlanView pv = view_plan(plan, ph);
        if (w >= pv.n_items) return;
        r = pv.item_row[w]; beg = pv.item_beg[w]; end = pv.item_end[w]; slot = pv.item_slot[w];
    } else {
        if (w >= N) return;
        r = w; beg = rowptr[r]; end = rowptr[r + 1];
    }
    if (SPARSE == kSparseOut && rows[r] == 0) return;
    const float4 acc = SPARSE == kSparseIn
                           ? gather_range_active<LPR>(col, val, X, rows, beg, end, sub, grp, s_ent + (threadIdx.x >> 6) * kWave)
                           : gather_range<LPR>(col, val, X, beg, end, sub, grp);
    if (grp == 0) {
        if (slot < 0) write_row<LPR>(r, sub, acc, Y, S_in, S_out, scale, SPARSE != kSparseIn || rows[r] != 0);
        else st4(slab + (size_t)slot * d + 4 * sub, acc);
    }
}

// Sums the chunk partials of every split (hub) row and applies the epilogue.  One wave per hub row:
// the 64/LPR groups take interleaved slots (two loads in flight each), then combine by shuffles --
// a fixed summation tree, so the result is deterministic.
template <int LPR, int SPARSE>
__global__ __launch_bounds__(256) void k_spmm_fixup(const void *__restrict__ plan, PlanHeader ph,
                                                    const float *__restrict__ slab, float *Y, const float *S_in,
                                                    float *S_out, float scale, const uint8_t *__restrict__ rows) {
    constexpr int d = 4 * LPR;
    constexpr int NG = kWave / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const PlanView pv = view_plan(plan, ph);
    if (k >= pv.n_split) return;
    if (SPARSE == kSparseOut && rows[pv.split_row[k]] == 0) return;
    const int s0 = pv.split_slot0[k], s1 = pv.split_slot0[k + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
    int s = s0 + grp;
    for (; s + NG < s1; s += 2 * NG) {
        acc = add4(acc, ld4(slab + (size_t)s * d + 4 * sub));
        acc2 = add4(acc2, ld4(slab + (size_t)(s + NG) * d + 4 * sub));
    }
    if (s < s1) acc = add4(acc, ld4(slab + (size_t)s * d + 4 * sub));
    acc = add4(acc, acc2);
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
    }
    if (grp == 0) write_row<LPR>(pv.split_row[k], sub, acc, Y, S_in, S_out, scale, SPARSE != kSparseIn || rows[pv.split_row[k]] != 0);
}

// out = in * scale   (n_layers == 0 degenerate case) -- float4 per lane
__global__ void k_scale_copy(size_t n_vec, const float *__restrict__ in, float *__restrict__ out, float scale) {
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * blockDim.x) {
        const float4 x = ld4(in + v * 4);
        st4(out + v * 4, make_float4(x.x * scale, x.y * scale, x.z * scale, x.w * scale));
    }
}

// work: [2*N*d] layer buffers followed by [n_slots*d] slab when a plan is given.
// sparse_rows (may be NULL) with mode kSparseOut: only the flagged rows of E are wanted (computed in the last layer);
// with mode kSparseIn: E0 is row-sparse, only its flagged rows are non-zero (and only they are read).
int launch_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col, const float *val,
                     const void *plan_dev, const void *plan_host_header, const float *E0, float *E, float *work,
                     hipStream_t st, const uint8_t *sparse_rows, int sparse_mode) {
    const size_t nd = (size_t)N * d;
    float *bufA = work, *bufB = work + nd;            // alternating layer outputs
    float *slab = work + 2 * nd;
    const float inv = 1.0f / (float)(n_layers + 1);
    if (n_layers == 0) {
        k_scale_copy<<<1024, 256, 0, st>>>(nd / 4, E0, E, 1.0f);
        MACR_CHECK_LAUNCH("scale_copy", st);
        return MACR_OK;
    }
    PlanHeader ph = {};
    if (plan_dev) ph = *static_cast<const PlanHeader *>(plan_host_header);
    const int n_waves = plan_dev ? ph.n_items : N;
    const int grid = (n_waves + 3) / 4;
    const float *X = E0;
    const float *S_in = E0;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        float *Y = last ? nullptr : ((l & 1) ? bufB : bufA);
        const float scale = last ? inv : 1.0f;        // running sum lives in E; first layer reads E0 as S_in
        const int mode = !sparse_rows ? kDense
                         : (sparse_mode == kSparseOut && last) ? kSparseOut
                         : (sparse_mode == kSparseIn && l == 0) ? kSparseIn : kDense;
        const int n_fix = (ph.n_split + 3) / 4;
        if (mode == kSparseOut) {
            MACR_DISPATCH_LPR(d, (k_spmm_csr<LPR, kSparseOut><<<grid, 256, 0, st>>>(N, rowptr, col, val, plan_dev, ph, X, Y, S_in, E,
                                                                                    scale, slab, sparse_rows)));
            MACR_CHECK_LAUNCH("spmm_csr_rows", st);
            if (plan_dev && ph.n_split > 0) {
                MACR_DISPATCH_LPR(d, (k_spmm_fixup<LPR, kSparseOut><<<n_fix, 256, 0, st>>>(plan_dev, ph, slab, Y, S_in, E, scale, sparse_rows)));
                MACR_CHECK_LAUNCH("spmm_fixup", st);
            }
        } else if (mode == kSparseIn) {
            MACR_DISPATCH_LPR(d, (k_spmm_csr<LPR, kSparseIn><<<grid, 256, 0, st>>>(N, rowptr, col, val, plan_dev, ph, X, Y, S_in, E,
                                                                                   scale, slab, sparse_rows)));
            MACR_CHECK_LAUNCH("spmm_csr_sparse", st);
            if (plan_dev && ph.n_split > 0) {
                MACR_DISPATCH_LPR(d, (k_spmm_fixup<LPR, kSparseIn><<<n_fix, 256, 0, st>>>(plan_dev, ph, slab, Y, S_in, E, scale, sparse_rows)));
                MACR_CHECK_LAUNCH("spmm_fixup", st);
            }
        } else {
            MACR_DISPATCH_LPR(d, (k_spmm_csr<LPR, kDense><<<grid, 256, 0, st>>>(N, rowptr, col, val, plan_dev, ph, X, Y, S_in, E,
                                                                                scale, slab, nullptr)));
            MACR_CHECK_LAUNCH("spmm_csr", st);
            if (plan_dev && ph.n_split > 0) {
                MACR_DISPATCH_LPR(d, (k_spmm_fixup<LPR, kDense><<<n_fix, 256, 0, st>>>(plan_dev, ph, slab, Y, S_in, E, scale, nullptr)));
                MACR_CHECK_LAUNCH("spmm_fixup", st);
            }
        }
        X = Y;
        S_in = E;
    }
    return MACR_OK;
}

static void build_plan(int N, const int32_t *rowptr, std::vector<int32_t> &out) {
    std::vector<int32_t> irow, ibeg, iend, islot, srow, sslot0;
    int n_slots = 0;
    for (int r = 0; r < N; ++r) {
        const int beg = rowptr[r], end = rowptr[r + 1];
        if (end - beg <= kChunk) {
            irow.push_back(r); ibeg.push_back(beg); iend.push_back(end); islot.push_back(-1);
        } else {
            srow.push_back(r); sslot0.push_back(n_slots);
            for (int b = beg; b < end; b += kChunk) {
                irow.push_back(r); ibeg.push_back(b); iend.push_back(b + kChunk < end ? b + kChunk : end);
                islot.push_back(n_slots++);
            }
        }
    }
    sslot0.push_back(n_slots);
    PlanHeader h = {};
    h.magic = kPlanMagic; h.n_items = (int32_t)irow.size(); h.n_split = (int32_t)srow.size(); h.n_slots = n_slots; h.N = N;
    out.assign(reinterpret_cast<int32_t *>(&h), reinterpret_cast<int32_t *>(&h) + sizeof(h) / 4);
    out.insert(out.end(), irow.begin(), irow.end());
    out.insert(out.end(), ibeg.begin(), ibeg.end());
    out.insert(out.end(), iend.begin(), iend.end());
    out.insert(out.end(), islot.begin(), islot.end());
    out.insert(out.end(), srow.begin(), srow.end());
    out.insert(out.end(), sslot0.begin(), sslot0.end());
}

}  // namespace macr

using namespace macr;

// ---- plan (host) ----------------------------------------------------------------
extern "C" size_t macr_spmm_plan_bytes(int N, const int32_t *rowptr_host) {
    if (N <= 0 || !rowptr_host) return 0;
    size_t items = 0, split = 0;
    for (int r = 0; r < N; ++r) {
        const int len = rowptr_host[r + 1] - rowptr_host[r];
        if (len <= kChunk) items += 1; else { items += (len + kChunk - 1) / kChunk; split += 1; }
    }
    return sizeof(PlanHeader) + 4 * (4 * items + split + split + 1);
}

extern "C" int macr_spmm_plan_build(int N, const int32_t *rowptr_host, void *plan_host, size_t plan_bytes) {
    MACR_REQUIRE(N > 0 && rowptr_host && plan_host, MACR_E_INVALID, "spmm_plan_build: bad arguments");
    std::vector<int32_t> v;
    build_plan(N, rowptr_host, v);
    MACR_REQUIRE(plan_bytes >= v.size() * 4, MACR_E_WORKSPACE, "spmm_plan_build: buffer %zu < %zu bytes", plan_bytes, v.size() * 4);
    memcpy(plan_host, v.data(), v.size() * 4);
    return MACR_OK;
}

extern "C" size_t macr_lgcn_work_floats(int N, int d, const void *plan_host) {
    size_t n = (size_t)2 * N * d;
    if (plan_host) n += (size_t)static_cast<const PlanHeader *>(plan_host)->n_slots * d;
    return n;
}

static int check_plan(const void *plan_dev, const void *plan_host, int N, const char *who) {
    MACR_REQUIRE((plan_dev == nullptr) == (plan_host == nullptr), MACR_E_INVALID,
                 "%s: plan needs both its device copy and its host copy (or neither)", who);
    if (plan_host) {
        const PlanHeader *h = static_cast<const PlanHeader *>(plan_host);
        MACR_REQUIRE(h->magic == kPlanMagic && h->N == N, MACR_E_INVALID, "%s: plan does not belong to this graph", who);
    }
    return MACR_OK;
}

extern "C" int macr_lgcn_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col,
                                   const float *val, const void *plan_dev, const void *plan_host, const float *E0,
                                   float *E, float *work, void *stream) {
    MACR_REQUIRE(N > 0 && n_layers >= 0, MACR_E_INVALID, "lgcn_propagate: N=%d n_layers=%d", N, n_layers);
    MACR_REQUIRE(macr::dim_supported(d), MACR_E_UNSUPPORTED, "lgcn_propagate: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(rowptr && col && val && E0 && E && (work || (n_layers < 2 && !plan_dev)), MACR_E_INVALID,
                 "lgcn_propagate: null pointer");
    MACR_REQUIRE(E != E0, MACR_E_INVALID, "lgcn_propagate: E must not alias E0");
    if (int e = check_plan(plan_dev, plan_host, N, "lgcn_propagate")) return e;
    return macr::launch_propagate(N, d, n_layers, rowptr, col, val, plan_dev, plan_host, E0, E, work,
                                  macr::as_stream(stream), nullptr, 0);
}
