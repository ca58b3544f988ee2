// LightGCN normalised-adjacency SpMM for gfx950.
//
// Replaces the tf.sparse_tensor_dense_matmul loop of
// macr_lightgcn/LightGCN.py:297-305 (100 row folds x n_layers) and the
// stack/reduce_mean of :306-307.  A_hat = D^-1/2 A D^-1/2 in CSR
// (macr_lightgcn/utility/load_data.py:112-121), X is the (N,d) embedding table.
//
// THE WAVE IS THE ROW.  Lane l owns column l of the output row (columns l, l+64, ... at d >= 128; at d = 32 the two
// half-waves take alternate neighbours and are added at the end).  A neighbour row is then ONE coalesced dword load
// per lane whose base address is wave-uniform: column index and weight come out of the chunk's index registers
// (lane l <- entry base+l, one coalesced load each) with v_readlane, the address is scalar arithmetic, the product is
// a v_fmac with a scalar weight.  No cross-lane shuffles, no reduction tree, a row pads to a multiple of 8 entries.
//
// What a gather costs on this chip (tools/ta_bench.hip, every CU loaded, L1/L2-resident rows): a wave-level
// `global_load_dwordx4` takes ~16 cycles of its CU's vector memory pipe HOWEVER MANY LANES ARE ENABLED, a
// `global_load_dword` (64 lanes x 4 B = one 256-byte row) ~5 cycles; rows that miss the XCD's 4 MB L2 arrive at
// ~10 TB/s chip-wide (17 cycles per 256-byte row and CU) whatever the instruction.  The first two versions of this
// file gathered float4 per lane, four rows per instruction, in rounds of 16 entries: 857 k wave loads for the 683 k
// the Yelp2018-shape graph needs (every row pads to a multiple of 16), plus ~78 pipe cycles per work item for its
// descriptor, index, S_in and 16-lane float4 epilogue accesses.  Measured on the way (DESIGN.md section 4 has the table):
// redirecting every gather to 64 L1-resident rows took a layer from 60 to 46 us -- the kernel sits on the vector
// memory pipe, and the misses (29 % of the gathered rows at a 17.8 MB table) cost as much as all hits together; pinning
// the two sides of the bipartite graph to XCD halves, non-temporal loads for cold rows / index streams, hub pieces
// ordered by source range, and a persistent variant that kept two gather rounds in flight across row boundaries
// all lowered FETCH_SIZE or the trip count and were slower or equal in time.
//
// Work items come from a static plan (the graph never changes): a row with at most kChunk non-zeros is one item; a
// longer row (interaction graphs have hub items with 10^4..10^5 neighbours -- Addressa item 0 has 7077, the Zipf
// synthetic Yelp-size graph 3*10^4) is cut into kChunk-sized PIECES whose partial rows go to a scratch slab; the piece
// that finishes LAST (one counter per hub row) sums the partials in slot order -- deterministic whichever piece that
// is -- and applies the epilogue.  (Until round 3 a fix-up kernel did that: four more launches per training step.)
// HBM-bound: compulsory bytes per layer nnz*8 + (N+1)*4 + 2*N*d*4.
#include "common.hpp"

#include <string.h>
#include <algorithm>
#include <vector>

namespace macr {

constexpr int kChunk = 512;       // non-zeros per work item (kernel-development knob: MACR_SPMM_CHUNK in the environment)

struct PlanHeader {               // all int32, followed by the arrays below
    int32_t magic, n_items, n_split, n_slots, N, chunk, n_groups, reserved;
    // RECORDS of the short rows (round 5, spmm_records below): item[0 .. n_single) are the pieces and the rows of more than
    // kRecEntries neighbours -- one wave each in a dense layer -- and every shorter row also sits in a record, several to a
    // wave: n_rec[c] records of class c (8, 4, 2, 1 rows of at most 4, 8, 16, 32 neighbours) from int32 offset rec_off on
    int32_t n_single, rec_off, n_rec[4], pad[2];
};
constexpr int32_t kPlanMagic = 0x4d414356;   // "MACV" (layout 5)
constexpr int kRecEntries = 32;              // neighbours per record
constexpr int kRecInts = 16 + 2 * kRecEntries;   // a record: 16 header ints (row ids, -1 = none) + 32 (column, weight) pairs = 320 B
constexpr int kGroup = 16;        // pieces per group
// layout after the header (64 bytes, so the descriptors are 16-byte aligned):
//   item[n_items] = {row, beg, end, slot}   the n_slots pieces of the split rows first (slot >= 0, in slot order), then
//                                           the other rows (slot = -1), longest first
//   slot_group[n_slots]                     the group a piece belongs to (kGroup consecutive pieces of one row)
//   group_slot0[n_groups+1]                 group g owns slots group_slot0[g] .. group_slot0[g+1]-1
//   group_split[n_groups]                   the hub row (index k) of a group
//   split_group0[n_split+1]                 hub row k owns groups split_group0[k] .. split_group0[k+1]-1
//   split_row[n_split]                      its row

struct PlanView {
    const int4 *item;
    const int32_t *slot_group, *group_slot0, *group_split, *split_group0, *split_row;
    int n_items, n_split, n_slots, n_groups;
};

__device__ __host__ inline PlanView view_plan(const void *plan, const PlanHeader &h) {
    const int32_t *p = reinterpret_cast<const int32_t *>(plan) + sizeof(PlanHeader) / 4;
    PlanView v;
    v.n_items = h.n_items; v.n_split = h.n_split; v.n_slots = h.n_slots; v.n_groups = h.n_groups;
    v.item = reinterpret_cast<const int4 *>(p); p += 4 * (size_t)h.n_items;
    v.slot_group = p; p += h.n_slots;
    v.group_slot0 = p; p += h.n_groups + 1;
    v.group_split = p; p += h.n_groups;
    v.split_group0 = p; p += h.n_split + 1;
    v.split_row = p;
    return v;
}

// SPARSE modes (a LightGCN training step only needs the propagated rows of its batch, and its gradient enters the
// backward propagation with <= 3B non-zero rows):
//   kSparseOut  only the rows the batch refers to (and every hub row) are computed (the LAST forward layer: nothing
//               else is read afterwards).  One wave per batch reference -- a row referred to twice is computed twice,
//               to the same values; the hub rows, which hold the repeats of a popularity-skewed batch, are left to
//               their pieces.  (Building a duplicate-free row list first costs more than it saves: the atomics that
//               detect a row's first reference queue on the hot items' cache lines, ~24 ns each: 45 us per batch.)
//   kSparseIn   X and S_in are row-sparse: only rows with rows[.] != 0 are read (the FIRST backward layer)
constexpr int kDense = 0, kSparseOut = 1, kSparseIn = 2;

struct SparseCtx {                // device pointers
    int32_t *cnt;                 // [N] how often the current batch refers to a row (0: not a row of the batch).  The last
                                  // forward layer COUNTS (one wave per reference), the first backward layer reads the
                                  // counts as flags, the fused epilogue of the last backward layer consumes and clears them
    const int32_t *u, *i, *j;     // the batch: rows u[b], n_users + i[b], n_users + j[b]
    int B, n_users;
    int chunk;                    // rows longer than this are hub rows (cut into pieces by the plan); INT_MAX without a plan
    int count;                    // kSparseOut: 1 = count the references into cnt (a training step), 0 = leave it alone
};

// The optimizer, fused into the epilogue of the LAST backward layer (a LightGCN training step; LightGCN.py:186/:201 are
// dense Adam over the ego table T): the wave that finishes row r holds its gradient (S_in[r] + A X[r]) * scale in
// registers and applies it on the spot -- together with the ego-row regulariser coef * cnt[r] * T[r] (LightGCN.py:525-528:
// one term per reference) and the row's share cnt[r] * |T[r]|^2 of emb_loss -- instead of writing G for a separate pass
// over the table (125 MB of traffic and two more launches per step).  It also clears cnt[r] and the row dE[r] of the
// gradient buffer the pair kernels accumulate into: both are zero again when the next step starts.
struct AdamFuse {
    float *T, *m, *v;             // ego table and its Adam slots, updated in place (NULL: no fusion)
    const StepScalars *scal;      // lr_t of this step (written by pair_bwd)
    float b1, b2, eps, coef;      // coef = decay / batch_size
    float *dE;                    // gradient buffer of the pair kernels (rows of the batch are zeroed)
    double *emb_acc;              // [2048] partial sums of cnt * |T row|^2 (emb_loss), slot = row % 2048
};

struct SpmmArgs {
    int N, n_items, n_slots;
    const int32_t *rowptr, *col;
    const float *val;
    const int4 *items;            // NULL: no plan, item w = row w
    const int32_t *slot_group, *group_slot0, *group_split, *split_group0;
    int n_groups;
    int32_t *arrivals;            // [n_groups + n_split] arrival counters of the groups, then of the hub rows; zero between
                                  // launches (whoever arrives last resets the counter)
    const float *X;
    float *Y;
    const float *S_in;
    float *S_out;
    float scale;
    float *slab;                  // [n_slots + n_groups][d] partial rows of the pieces, then of the groups
    SparseCtx sp;
};

// Element (row c, column k) of a table below 4 GB (checked by the launcher): wave-uniform base + 32-bit byte offset, i.e.
// one VALU instruction per gathered row (`v_lshl_add_u32` + `global_load_dword v, v_off, s[base]`) and no scalar one --
// a CU has ONE scalar unit for its four SIMDs, and 64-bit address arithmetic per neighbour kept it busy.
template <int D>
__device__ __forceinline__ float ld_elem(const float *__restrict__ X, int c, int k) {
#ifdef MACR_ABL_SPMM_ROWMASK                                      // timing probe (wrong results): gather from the first few rows only
    c &= MACR_ABL_SPMM_ROWMASK;
#endif
    const uint32_t off = ((uint32_t)c * (uint32_t)D + (uint32_t)k) * 4u;
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + off);
}
#ifdef MACR_ABL_SPMM_LDS                                          // timing probe (wrong results): gathers from LDS rows
__shared__ float abl_lds[128 * 64];
template <int D>
__device__ __forceinline__ float ld_elem_lds(int c, int k) { return abl_lds[(c & 127) * 64 + (k & 63)]; }
#endif

// the same with the row index in a VGPR (wave-uniform all the same: it came out of LDS) and the lane's byte offset given
template <int D>
__device__ __forceinline__ float ld_elem_v(const float *__restrict__ X, int c, uint32_t lane_bytes) {
#ifdef MACR_ABL_SPMM_ROWMASK
    c &= MACR_ABL_SPMM_ROWMASK;
#endif
    const uint32_t off = (uint32_t)c * (uint32_t)(D * 4) + lane_bytes;     // (an end marker's bit 31 leaves with the overflow)
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + off);
}

__device__ __forceinline__ bool row_flag(const int32_t *cnt, int r) { return cnt[r] != 0; }

#ifdef MACR_SPMM_TRACE                                            // kernel-development probe (tools/spmm_bench.hip): per-wave phase stamps
__device__ unsigned long long g_spmm_trace[1 << 17][8];
#define SPMM_STAMP(w_, k_) do { if (lane == 0 && (w_) < (1 << 17)) g_spmm_trace[(w_)][(k_)] = __builtin_readcyclecounter(); } while (0)
#define SPMM_STAMP_RT(w_, k_) do { if (lane == 0 && (w_) < (1 << 17)) g_spmm_trace[(w_)][(k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SPMM_STAMP(w_, k_) do {} while (0)
#define SPMM_STAMP_RT(w_, k_) do {} while (0)
#endif

// Lane geometry: D >= 64: lane l holds columns l + 64 v (v < NV), one entry per step; D = 32: lane l holds column
// l % 32, the half-waves take alternate entries (two per step).
template <int D> struct RowGeom {
    static constexpr bool kHalf = D == 32;
    static constexpr int NV = kHalf ? 1 : D / 64;
    static constexpr int EPS = kHalf ? 2 : 1;       // entries per step
#ifndef MACR_SPMM_MAXNB
#define MACR_SPMM_MAXNB 4
#endif
    static constexpr int MAXNB = NV >= 4 ? 1 : NV == 2 ? 2 : MACR_SPMM_MAXNB;   // batches in flight: at most 32 row registers
};

// NB batches of 8 steps starting at entry `pos` of the CSR arrays, all loads issued before the first use; entries
// at and behind `left` get weight 0 (their columns belong to the next row: valid rows).  Column indices and weights
// arrive by SCALAR loads (pos is wave-uniform) -- straight into the registers the address arithmetic and the fma
// take them from.  The version before took them out of the lanes of a coalesced vector load with v_readlane: 4 VALU
// instructions per neighbour, two of them 8-cycle v_readlanes -- PMC: the VALU was busy 82 % of the layer's time, and
// neither L1-resident rows nor shorter hub chains changed its 54 us.
// The batch count is a template parameter: a batch under a run-time guard makes its registers merge with the skipped
// path, and the compiler resolves such a merge by waiting for the loads on the spot.
// SAFE: the window may reach behind the end of the arrays (the last rows of the matrix): every index is clamped.
template <int D, int NB, bool SAFE = false>
__device__ __forceinline__ void gather_batches(const int32_t *__restrict__ col, const float *__restrict__ val, int pos,
                                               int left, int last, const float *__restrict__ X, int lane,
                                               float (&acc)[RowGeom<D>::NV]) {
    using G = RowGeom<D>;
    constexpr int NE = NB * 8 * G::EPS;                          // entries
    int c[NE]; float a[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int q = SAFE ? (pos + e < last ? pos + e : last) : pos + e;
        c[e] = col[q]; a[e] = val[q];
    }
    // weights of entries at and behind `left` are dropped -- only the window's last batch can hold any; the select works
    // on the bit patterns so that it stays a scalar instruction
    auto wgt = [&](int e) {
        if (e < NE - 8 * G::EPS) return a[e];
        return __builtin_bit_cast(float, e < left ? __builtin_bit_cast(int, a[e]) : 0);
    };
    float x[NB * 8][G::NV]; float w[NB * 8];
#pragma unroll
    for (int k = 0; k < NB * 8; ++k) {
        if (G::kHalf) {
            const float a0 = wgt(2 * k), a1 = wgt(2 * k + 1);
            w[k] = lane < 32 ? a0 : a1;
            x[k][0] = ld_elem<D>(X, lane < 32 ? c[2 * k] : c[2 * k + 1], lane & 31);
        } else {
            w[k] = wgt(k);                                       // wave-uniform
#pragma unroll
            for (int v = 0; v < G::NV; ++v) {
#ifdef MACR_ABL_SPMM_LDS
                if (MACR_ABL_SPMM_LDS == 1 || (k & 1)) { x[k][v] = ld_elem_lds<D>(c[k], lane + 64 * v); continue; }
#endif
                x[k][v] = ld_elem<D>(X, c[k], lane + 64 * v);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB * 8; ++k) {
#pragma unroll
        for (int v = 0; v < G::NV; ++v) acc[v] = fmaf(w[k], x[k][v], acc[v]);
    }
}


// The same for a ROW-SPARSE X: `m` marks the chunk's entries whose source row is active; the steps walk its set bits in
// ascending order (when the mask runs out, the last entry is repeated with weight 0).
template <int D, int NB>
__device__ __forceinline__ void gather_batches_masked(int cv, float av, uint64_t &m, const float *__restrict__ X, int lane,
                                                      float (&acc)[RowGeom<D>::NV]) {
    using G = RowGeom<D>;
    float x[NB * 8][G::NV]; float a[NB * 8];
    int idx = 0;
    auto next = [&](int &c, float &w) {                          // wave-uniform
        const bool on = m != 0ull;
        idx = on ? __builtin_ctzll(m) : idx;
        m &= m - 1ull;
        c = __builtin_amdgcn_readlane(cv, idx);
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, av), idx));
        w = on ? t : 0.f;
    };
#pragma unroll
    for (int k = 0; k < NB * 8; ++k) {
        if (G::kHalf) {
            int c0, c1; float a0, a1;
            next(c0, a0); next(c1, a1);
            const int c = lane < 32 ? c0 : c1;
            a[k] = lane < 32 ? a0 : a1;
            x[k][0] = ld_elem<D>(X, c, lane & 31);
        } else {
            int c;
            next(c, a[k]);
#pragma unroll
            for (int v = 0; v < G::NV; ++v) x[k][v] = ld_elem<D>(X, c, lane + 64 * v);
        }
    }
#pragma unroll
    for (int k = 0; k < NB * 8; ++k) {
#pragma unroll
        for (int v = 0; v < G::NV; ++v) acc[v] = fmaf(a[k], x[k][v], acc[v]);
    }
}


// RECORDS (round 5): the short rows of a dense layer.
// Per-wave phase stamps of a Yelp2018-shape layer (tools/spmm_bench.hip -DMACR_SPMM_TRACE, profiles/r05_spmm_trace_*.txt): 51 k
// of the layer's 73 k waves are rows of <= 32 neighbours; such a wave lives ~3 900 cycles -- descriptor 850, index window and
// gathers 2 000, store 1 000 -- for 2 to 19 gathers, the chip holds ~5 800 of them at a time whatever the workgroup size, and
// the second half of the launch is that: waves being started to walk a chain of dependent trips while the vector memory pipe
// idles.  Several short rows per wave with the CSR arrays did not help (each row's window is a scalar load of its own, a miss
// costs ~800 cycles and a wave's misses are served one after the other: 6 800 cycles for eight windows).  So the plan owns a
// packed copy of the short rows: a RECORD is 16 header ints (row ids) + 32 (column, weight) pairs, contiguous, for R = 8 / 4 /
// 2 / 1 rows of at most 32 / R neighbours (padding weighs 0).  A wave knows its record from its index alone: ONE coalesced
// vector load brings rows, columns and weights (no descriptor trip, no scalar loads), LDS hands every lane the pair of step k
// (broadcast ds_read_b64), the 32 gathers and the rows' S_in / optimizer operands fly together, R epilogues follow.  Two
// dependent trips for up to eight rows instead of three per row; 20 k waves instead of 51 k.  Same arithmetic per row (CSR
// order, one fma chain).  6.6 MB of plan for the Yelp2018-shape graph.
template <int D, int R, bool FUSE>
__device__ __forceinline__ void spmm_records(const int32_t *__restrict__ rec, const float *__restrict__ X, float *Y, const float *S_in,
                                             float *S_out, float scale, int32_t *sp_rows, float *fT, float *fm, float *fv,
                                             const StepScalars *__restrict__ fscal, float *fdE, double *femb, float b1, float b2,
                                             float eps, float coef, int w_trace = 0) {
    static_assert(D == 64, "records: one column per lane");
#ifdef MACR_SPMM_TRACE
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime(), tr_c0 = __builtin_readcyclecounter();
#endif
    constexpr int E = kRecEntries / R;                           // neighbours per row
    __shared__ uint2 s_rec[4][64];
    const int lane = threadIdx.x & 63;
    uint2 *win = s_rec[threadIdx.x >> 6];
    // lanes 0..31: pair l; lanes 32..39: header ints 2 (l - 32), 2 (l - 32) + 1 (row ids)
    const uint2 mine = lane < 32 ? reinterpret_cast<const uint2 *>(rec + 16)[lane]
                                 : (lane < 40 ? reinterpret_cast<const uint2 *>(rec)[lane - 32] : make_uint2(0u, 0u));
    win[lane] = mine;
    int row[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const uint2 h = win[32 + (q >> 1)];                      // wave-uniform address: a broadcast read
        row[q] = __builtin_amdgcn_readfirstlane((int)((q & 1) ? h.y : h.x));
    }
#ifdef MACR_SPMM_TRACE
    SPMM_STAMP(w_trace, 2);                                       // the record has arrived
    if (lane == 0 && w_trace < (1 << 17)) { g_spmm_trace[w_trace][0] = tr_rt0; g_spmm_trace[w_trace][1] = tr_c0; g_spmm_trace[w_trace][3] = 0;
        g_spmm_trace[w_trace][7] = ((unsigned long long)(unsigned)kRecEntries << 32) | (unsigned)(R << 1); }
#endif
    float s[R], th0[R], m0[R], v0[R];
    int c0[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const size_t o = (size_t)(row[q] < 0 ? 0 : row[q]) * D + lane;
        s[q] = (S_out != nullptr || FUSE) ? S_in[o] : 0.f;
        if (FUSE) { c0[q] = sp_rows[row[q] < 0 ? 0 : row[q]]; th0[q] = fT[o]; m0[q] = fm[o]; v0[q] = fv[o]; }
    }
    float x[kRecEntries];
    const uint32_t lane_bytes = (uint32_t)lane * 4u;
#pragma unroll
    for (int k = 0; k < kRecEntries; ++k) {
        const uint32_t cb = win[k].x * (uint32_t)(D * 4);
        x[k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + (cb + lane_bytes));
    }
#ifdef MACR_SPMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPMM_STAMP(w_trace, 4);
#endif
#pragma unroll
    for (int q = 0; q < R; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) acc = fmaf(__builtin_bit_cast(float, win[q * E + e].y), x[q * E + e], acc);
        if (row[q] < 0) continue;                                // (wave-uniform: padding of the last record of a class)
        const size_t o = (size_t)row[q] * D + lane;
        if (FUSE) {
            const int cr = __builtin_amdgcn_readfirstlane(c0[q]);
            float th = th0[q], m = m0[q], vv = v0[q], sq = 0.f;
            float g = (s[q] + acc) * scale;
            if (cr) { g = fmaf(coef * (float)cr, th, g); sq = th * th; fdE[o] = 0.f; }
            adam1(th, m, vv, g, fscal->lr_t, b1, b2, eps);
            fT[o] = th; fm[o] = m; fv[o] = vv;
            if (cr) {
                sq = wave_sum(sq);
                if (lane == 0) { atomicAdd(femb + (row[q] & 2047), (double)cr * (double)sq); sp_rows[row[q]] = 0; }
            }
        } else {
            if (Y) Y[o] = acc;
            if (S_out) S_out[o] = (s[q] + acc) * scale;
        }
    }
#ifdef MACR_SPMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPMM_STAMP(w_trace, 5); SPMM_STAMP_RT(w_trace, 6);
#endif
}

// Y = A X (if Y), S_out = (S_in + A X) * scale (if S_out).  S_out may alias S_in in the dense modes (a row is read and
// written by its own wave only); in kSparseOut mode a row may be computed by several waves, so it must not.
// Every read-only array is a kernel parameter of its own with __restrict__: pointers that arrive inside a struct carry no
// no-alias guarantee, and a load the compiler cannot prove unclobbered by the stores of an earlier loop iteration is
// no longer a scalar load but a vector load + v_readfirstlane with a wait right behind it (measured: 61 -> 98 us).
struct SpmmScalars {
    int N, n_items, n_slots, n_groups;
    float scale;
    int B, n_users, chunk, count; // SparseCtx
    float b1, b2, eps, coef;      // AdamFuse
    int n_single, n_rec8, n_rec4, n_rec2, n_rec1;   // RECORDS (dense layers at d = 64; n_single = 0: off): waves [0, n_single) take one
                                  // item each, the waves behind them one record each, class by class
};
template <int D, int SPARSE, bool FUSE>
__global__ __launch_bounds__(256) void k_spmm_row(const SpmmScalars P, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const float *__restrict__ val,
                                                  const int4 *__restrict__ items, const int32_t *__restrict__ slot_group,
                                                  const int32_t *__restrict__ group_slot0,
                                                  const int32_t *__restrict__ group_split,
                                                  const int32_t *__restrict__ split_group0, int32_t *arrivals,
                                                  const float *__restrict__ X, float *Y, const float *S_in, float *S_out,
                                                  float *slab, int32_t *sp_cnt,
                                                  const int32_t *__restrict__ sp_u, const int32_t *__restrict__ sp_i,
                                                  const int32_t *__restrict__ sp_j, float *fT, float *fm, float *fv,
                                                  const StepScalars *__restrict__ fscal, float *fdE, double *femb,
                                                  const int32_t *__restrict__ records) {
    struct {                       // (the names the body uses)
        int N, n_items, n_slots, n_groups; float scale;
        const int32_t *rowptr; const int4 *items;
        const int32_t *slot_group, *group_slot0, *group_split, *split_group0; int32_t *arrivals;
        float *Y; const float *S_in; float *S_out; float *slab;
        struct { int32_t *rows; const int32_t *u, *i, *j; int B, n_users, chunk, count; } sp;
    } A = {P.N, P.n_items, P.n_slots, P.n_groups, P.scale, rowptr, items, slot_group, group_slot0, group_split, split_group0,
           arrivals, Y, S_in, S_out, slab, {sp_cnt, sp_u, sp_i, sp_j, P.B, P.n_users, P.chunk, P.count}};
    using G = RowGeom<D>;
    constexpr int NV = G::NV;
    constexpr int EPB = 8 * G::EPS;                              // entries per batch
    constexpr int MAXNB = G::MAXNB;
    const int lane = threadIdx.x & 63;
    const int colofs = G::kHalf ? (lane & 31) : lane;
    const bool writer = !G::kHalf || lane < 32;
    // One wave per item.  (PERSISTENT waves -- wave w takes items w, w + W, ... and asks for the next descriptor before it
    // works on the current one -- were measured twice and lost: 67-75 us against 61 for a Yelp2018-shape layer.  With
    // the gathers compiled out the layer still takes 29 of its 52 us: launch, arguments, descriptor, S_in, stores -- a
    // chain of ~3 dependent trips per item that a persistent wave walks just as serially.)
    const int total = SPARSE == kSparseOut ? A.n_slots + 3 * A.sp.B : A.items ? A.n_items : A.N;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };      // wave-uniform: scalar loads below
    // {row, beg, end, slot}; row = -1: nothing to do.  Each source loads inside its own branch and pins the values there
    // (empty asm): left alone, the compiler merges the branches into ONE load through a selected pointer, which has lost
    // its no-alias provenance and is then a vector load with a wait behind it.
    auto pin = [](int4 &d) { asm volatile("" : "+s"(d.x), "+s"(d.y), "+s"(d.z), "+s"(d.w)); };
    auto fetch = [&](int w) {
        int4 d;
        if (SPARSE == kSparseOut && w >= A.n_slots) {            // batch reference (the pieces of the hub rows come first)
            const int k = w - A.n_slots;
            const int which = k / A.sp.B, b = k - which * A.sp.B;
            const int r = which == 0 ? A.sp.u[b] : A.sp.n_users + (which == 1 ? A.sp.i[b] : A.sp.j[b]);
            // count the reference (no value returned: nothing waits).  Positives only when the pair kernels cannot
            // (count == 2): they repeat, and pair_bwd counts them with one atomic per distinct row of a chunk
            if ((A.sp.count == 2 || (A.sp.count == 1 && which != 1)) && lane == 0) atomicAdd(A.sp.rows + r, 1);
            const int beg = A.rowptr[r], end = A.rowptr[r + 1];
            d = make_int4(end - beg > A.sp.chunk ? -1 : r, beg, end, -1);     // a hub row: its pieces compute it
            pin(d);
        } else if (A.items) {
            d = A.items[w];
            pin(d);
        } else {
            d = make_int4(w, A.rowptr[w], A.rowptr[w + 1], -1);
            pin(d);
        }
        return d;
    };
    const int w = uni(blockIdx.x * 4 + (threadIdx.x >> 6));
    if constexpr (SPARSE == kDense && D == 64) {
        if (P.n_single > 0 && w >= P.n_single) {                 // (wave-uniform) a record of short rows
            const int k = w - P.n_single;
            const int32_t *rec = records + (size_t)k * kRecInts;
#define MACR_REC_ARGS rec, X, Y, S_in, S_out, P.scale, sp_cnt, fT, fm, fv, fscal, fdE, femb, P.b1, P.b2, P.eps, P.coef, w
            if (k < P.n_rec8) spmm_records<D, 8, FUSE>(MACR_REC_ARGS);
            else if (k < P.n_rec8 + P.n_rec4) spmm_records<D, 4, FUSE>(MACR_REC_ARGS);
            else if (k < P.n_rec8 + P.n_rec4 + P.n_rec2) spmm_records<D, 2, FUSE>(MACR_REC_ARGS);
            else if (k < P.n_rec8 + P.n_rec4 + P.n_rec2 + P.n_rec1) spmm_records<D, 1, FUSE>(MACR_REC_ARGS);
#undef MACR_REC_ARGS
            return;
        }
    }
    if (w >= total) return;
#ifdef MACR_SPMM_TRACE
    const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime(), tr_c0 = __builtin_readcyclecounter();
#endif
    const int4 cur = fetch(w);
    const int r = cur.x, beg = cur.y, slot = cur.w;
    int end = cur.z;
    if (r < 0) return;
#ifdef MACR_SPMM_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SPMM_STAMP(w, 2);
    if (lane == 0 && w < (1 << 17)) { g_spmm_trace[w][0] = tr_rt0; g_spmm_trace[w][1] = tr_c0; g_spmm_trace[w][3] = 0;
        g_spmm_trace[w][7] = ((unsigned long long)(unsigned)(end - beg) << 32) | (unsigned)(slot >= 0); }
#endif
    const bool s_on = (A.S_out || FUSE) && (SPARSE != kSparseIn || row_flag(A.sp.rows, r));   // else S_in[r] counts as zero
    float s[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = (s_on && slot < 0) ? A.S_in[(size_t)r * D + colofs + 64 * v] : 0.f;   // used last, asked for first
    // fused optimizer: the row's parameter, its Adam slots and its reference count travel with the first gathers too
    float th0[NV], m0[NV], v0[NV];
    int c0 = 0;
    if (FUSE && slot < 0) {
        c0 = A.sp.rows[r];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const size_t o = (size_t)r * D + colofs + 64 * v;
            th0[v] = fT[o]; m0[v] = fm[o]; v0[v] = fv[o];
        }
    }
    float acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    if (SPARSE == kSparseIn) {
        // lane l <- column and weight of entry b+l of a chunk; padding lanes repeat the last entry (their weight is dropped)
        auto load_chunk = [&](int b, int &c, float &a) {
            const int m = end - b < kWave ? end - b : kWave;
            const int ec = b + (lane < m ? lane : m - 1);
            c = col[ec];
            a = val[ec];
        };
        int cv = 0; float a0 = 0.f;
        if (beg < end) {
            load_chunk(beg, cv, a0);
            // the first chunk's indices have to be HERE before the loop: the compiler places its wait for them inside
            // the loop, where it would also hold every later chunk until its own prefetch (issued just before) has arrived
            asm volatile("" : "+v"(cv), "+v"(a0));
        }
        for (int base = beg; base < end; base += kWave) {       // wave-uniform
            const int n = end - base < kWave ? end - base : kWave;
            int cvn = 0; float a0n = 0.f;
            if (base + kWave < end) load_chunk(base + kWave, cvn, a0n);   // the next chunk's indices travel with this chunk's rows
            const float av = lane < n ? a0 : 0.f;
            uint64_t m = __ballot(lane < n && row_flag(A.sp.rows, cv));
            while (m != 0ull) {                                 // wave-uniform
                const int left = __popcll(m);
                if (MAXNB >= 4 && left > 3 * EPB) gather_batches_masked<D, MAXNB >= 4 ? 4 : 1>(cv, av, m, X, lane, acc);
                else if (MAXNB >= 4 && left > 2 * EPB) gather_batches_masked<D, MAXNB >= 4 ? 3 : 1>(cv, av, m, X, lane, acc);
                else if (MAXNB >= 2 && left > EPB) gather_batches_masked<D, MAXNB >= 2 ? 2 : 1>(cv, av, m, X, lane, acc);
                else gather_batches_masked<D, 1>(cv, av, m, X, lane, acc);
            }
            cv = cvn; a0 = a0n;
        }
    } else {
        const int last = A.rowptr[A.N] - 1;                     // the arrays' last entry
#ifdef MACR_ABL_SPMM_NOGATHER
        end = beg;
#endif
        for (int pos = beg; pos < end; pos += MAXNB * EPB) {    // wave-uniform
            const int left = end - pos;
#ifdef MACR_SPMM_TRACE
            if (pos == beg + MAXNB * EPB) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SPMM_STAMP(w, 3); }   // first window done
#endif
            if (pos + MAXNB * EPB - 1 > last) {                 // (a window that could leave the arrays)
                for (int q = pos; q < end && q < pos + MAXNB * EPB; q += EPB)
                    gather_batches<D, 1, true>(col, val, q, end - q, last, X, lane, acc);
            }
            else if (MAXNB >= 4 && left > 3 * EPB) gather_batches<D, MAXNB >= 4 ? 4 : 1>(col, val, pos, left, last, X, lane, acc);
            else if (MAXNB >= 4 && left > 2 * EPB) gather_batches<D, MAXNB >= 4 ? 3 : 1>(col, val, pos, left, last, X, lane, acc);
            else if (MAXNB >= 2 && left > EPB) gather_batches<D, MAXNB >= 2 ? 2 : 1>(col, val, pos, left, last, X, lane, acc);
            else gather_batches<D, 1>(col, val, pos, left, last, X, lane, acc);
        }
    }
    if (G::kHalf) acc[0] += __shfl_xor(acc[0], 32, kWave);
#ifdef MACR_SPMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPMM_STAMP(w, 4);                                             // all gathers consumed
#endif
#ifdef MACR_ABL_SPMM_NOPIECE
    if (slot >= 0) return;
#endif
    if (slot >= 0) {
        // A piece of a hub row.  Partial rows go to the slab write-through (system-scope stores: past this XCD's L2), then
        // an arrival counter is bumped; whoever arrives LAST reads the partials (system-scope loads: not from its own
        // XCD's possibly stale L2) and sums them in slot order -- the same sum whichever wave that is.  Two levels: the
        // kGroup pieces of a group, then the groups of the row.  (One level over 512-entry pieces was the layer's
        // critical path: a wave gets 1/32 of its CU's memory pipe, and the 512 + 59 dependent loads of the largest
        // hub's last piece took ~60 us whatever the other 70 000 rows did.)
        auto publish = [&](int at) {
            if (writer) {
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    __hip_atomic_store(A.slab + (size_t)at * D + colofs + 64 * v, acc[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the partial is out before the arrival is counted
        };
        auto last_of = [&](int counter, int expected) {          // wave-uniform
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(A.arrivals + counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != expected - 1) return false;
            if (lane == 0) __hip_atomic_store(A.arrivals + counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
            return true;
        };
        auto sum_range = [&](int p0, int p1) {                   // acc = sum of slab rows [p0, p1), p1 - p0 <= kGroup
            float p[kGroup][NV];
#pragma unroll
            for (int q = 0; q < kGroup; ++q) {
                const int sq = p0 + q < p1 ? p0 + q : p1 - 1;
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    p[q][v] = __hip_atomic_load(A.slab + (size_t)sq * D + colofs + 64 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = 0.f;
#pragma unroll
            for (int q = 0; q < kGroup; ++q) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += p0 + q < p1 ? p[q][v] : 0.f;
            }
        };
        const int g = A.slot_group[slot];
        const int s0 = A.group_slot0[g], s1 = A.group_slot0[g + 1];
        const int k = A.group_split[g];
        const int g0 = A.split_group0[k], g1 = A.split_group0[k + 1];
        if (s1 - s0 > 1) {
            publish(slot);
            if (!last_of(g, s1 - s0)) return;
            sum_range(s0, s1);
        }
        if (g1 - g0 > 1) {
            publish(A.n_slots + g);
            if (!last_of(A.n_groups + k, g1 - g0)) return;
            for (int q0 = g0; q0 < g1; q0 += kGroup) {           // wave-uniform; more than kGroup groups: 256 * chunk non-zeros
                float part[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) part[v] = q0 == g0 ? 0.f : acc[v];
                sum_range(A.n_slots + q0, A.n_slots + (q0 + kGroup < g1 ? q0 + kGroup : g1));
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += part[v];
            }
        }
        if (s_on) {
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] = A.S_in[(size_t)r * D + colofs + 64 * v];
        }
    }
#ifdef MACR_ABL_SPMM_NOSTORE
    if (acc[0] != 123.f) return;
#endif
    if (FUSE) {
        // gradient of row r -> Adam on T[r] (see AdamFuse)
        if (slot >= 0) {                                         // the finisher of a hub row asks now
            c0 = A.sp.rows[r];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const size_t o = (size_t)r * D + colofs + 64 * v;
                th0[v] = fT[o]; m0[v] = fm[o]; v0[v] = fv[o];
            }
        }
        const int c = __builtin_amdgcn_readfirstlane(c0);       // references of the batch to this row (wave-uniform)
        const float lr_t = fscal->lr_t;
        float sq = 0.f;
        if (writer) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const size_t o = (size_t)r * D + colofs + 64 * v;
                float th = th0[v], m = m0[v], vv = v0[v];
                float g = (s[v] + acc[v]) * A.scale;
                if (c) { g = fmaf(P.coef * (float)c, th, g); sq = fmaf(th, th, sq); fdE[o] = 0.f; }
                adam1(th, m, vv, g, lr_t, P.b1, P.b2, P.eps);
                fT[o] = th; fm[o] = m; fv[o] = vv;
            }
        }
        if (c) {                                                 // wave-uniform
            sq = wave_sum(sq);
            if (lane == 0) {
                atomicAdd(femb + (r & 2047), (double)c * (double)sq);
                A.sp.rows[r] = 0;
            }
        }
        return;
    }
    if (writer) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const size_t o = (size_t)r * D + colofs + 64 * v;
            if (A.Y) A.Y[o] = acc[v];
            if (A.S_out) A.S_out[o] = (s[v] + acc[v]) * A.scale;
        }
    }
#ifdef MACR_SPMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPMM_STAMP(w, 5); SPMM_STAMP_RT(w, 6);
#endif
}

// =====================================================================================================================
// STREAM kernel (round 4): the dense layers.
//
// What the layer's time was made of (round 4 measurements, tools/spmm_bench.hip): with every gather served from LDS the
// layer still took 48 of its 51 us, with the gathers AND their index loads compiled out 14 -- the kernel was never short
// of gather bandwidth, it was a sum of chains: per wave  launch -> descriptor (cold) -> column/weight window (cold,
// scalar) -> gathers -> next window ... with nothing in flight while the next indices travel, 70 000 times.  Occupancy
// is at the hardware's 32 waves per CU, so the only levers are fewer, fatter work items and shorter chains:
//   * a work item is a CHUNK of a plan-owned entry stream (pcw: the matrix in row order as (column, weight) pairs, a few
//     hundred entries, whole rows only -- rows above kStreamHub entries are cut into pieces that are chunks of their
//     own); the wave walks it in rounds of 32 entries: 32 gathers in flight, 32 fmas;
//   * rows end INSIDE the stream: after a row's neighbours comes one more entry {row | bit 31, 0} whose gather reads the
//     row's S_in instead of X, always as the LAST entry of a group of 8 (padding in front of it) -- the wave then holds
//     (S_in + A X)[row] and stores; no descriptor, no row pointer, no per-row launch, one uniform branch per 8 entries;
//   * the stream does not touch the scalar caches (see the kernel): asynchronous LDS-DMA fills, broadcast LDS reads;
//   * rows without neighbours never enter the stream: each chunk takes its share of a list of them.
// =====================================================================================================================
struct StreamHeader {             // all int32; follows the row plan at PlanHeader.reserved (in int32 units)
    int32_t magic, n_chunks, n_sb, n_empty, n_slots, n_groups, n_split, n_entries;
};
constexpr int32_t kStreamMagic = 0x4d535452;   // "MSTR"
constexpr int kStreamHubDefault = 512;   // rows with more neighbours are cut into pieces (MACR_SPMM_HUB at plan time)
constexpr int kStreamPiece = 256; // entries per piece (255 neighbours + its end marker)
// layout after the header: chunk_desc[n_chunks] = {first sub-batch (32 entries), end, first empty row | count << 24, slot or -1}, empties[n_empty],
// slot_group[n_slots], group_slot0[n_groups+1], group_split[n_groups], split_group0[n_split+1],
// split_row[n_split], then (64-byte aligned) pcw[n_entries + 128] = {column | bit 31 for an end marker, weight bits (marker: 0 row, 1 piece)}
struct StreamView {
    const int4 *chunk_desc;
    const int32_t *empties;
    const int32_t *slot_group, *group_slot0, *group_split, *split_group0, *split_row;
    const int2 *pcw;
};
// plan: the start of the whole plan buffer (host or device copy), off: PlanHeader.reserved.  The arrays sit at offsets
// that are multiples of 64 bytes from the START of the buffer (the device copy is allocated 256-byte aligned).
__host__ __device__ inline StreamView view_stream(const void *plan, int32_t off, const StreamHeader &h) {
    const int32_t *base = reinterpret_cast<const int32_t *>(plan);
    size_t o = (size_t)off + sizeof(StreamHeader) / 4;
    o = (o + 3) / 4 * 4;                                           // chunk descriptors 16-byte aligned
    auto take = [&](size_t n) { const int32_t *p = base + o; o += n; return p; };
    StreamView v;
    v.chunk_desc = reinterpret_cast<const int4 *>(take(4 * (size_t)h.n_chunks));
    v.empties = take(h.n_empty);
    v.slot_group = take(h.n_slots);
    v.group_slot0 = take(h.n_groups + 1);
    v.group_split = take(h.n_groups);
    v.split_group0 = take(h.n_split + 1);
    v.split_row = take(h.n_split);
    o = (o + 15) / 16 * 16;
    v.pcw = reinterpret_cast<const int2 *>(take(2 * ((size_t)h.n_entries + 128)));
    return v;
}


// An end marker's gather must not cost a select per entry: it reads S_in through X's base register with s_in_bytes added
// to its offset, where S_in = X + s_in_bytes -- 0 for the first layer (S_in is X), N rows for the others, whose buffers
// the launcher keeps as pairs [X | S_in].  Only a group's last entry can be a marker (bit 31 of its index).
// The kernel takes ONE struct: what the rounds need (hot) is read from the argument as usual and lives in SGPRs; what only a
// flush or the hub reduction needs (cold) is fetched from the kernel-argument segment on the spot -- otherwise the compiler
// keeps all of it in scalar registers for the whole kernel and spills the operands of the rounds.
struct StreamHub { const int32_t *plan; int32_t off; int32_t pad; int32_t *arrivals; float *slab; };
struct StreamFuse { int32_t *cnt; float *T, *m, *v; const StepScalars *scal; float *dE; double *emb; };
struct StreamArgs {
    // hot
    const int4 *chunk_desc; const int32_t *empties; const int2 *pcw; uint32_t s_in_bytes; uint32_t pad0;
    const float *X;               // end markers gather S_in through X too:
    const float *S_in;            // S_in == X + s_in_bytes (the launcher lays the layer buffers out that way, or passes X itself)
    int n_chunks; float scale;
    // cold
    float *Y; float *S_out;
    float b1, b2, eps, coef;
    StreamHub hub; StreamFuse fz;
};

// One row's result: Y = A X, S_out = (S_in + A X) * scale -- or, FUSE, the optimizer on T[r] (see AdamFuse).
template <int D, bool FUSE>
__device__ __forceinline__ void stream_row_out(int r, const float (&acc)[RowGeom<D>::NV], const float (&s)[RowGeom<D>::NV],
                                               int lane, float scale) {
    constexpr int NV = RowGeom<D>::NV;
    const auto *ka = (const __attribute__((address_space(4))) StreamArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    if (FUSE) {
        int32_t *cnt = ka->fz.cnt;
        float *fT = ka->fz.T, *fm = ka->fz.m, *fv = ka->fz.v, *fdE = ka->fz.dE;
        const int c = cnt[r];                                    // references of the batch to this row (wave-uniform)
        const float lr_t = ka->fz.scal->lr_t;
        const float b1 = ka->b1, b2 = ka->b2, eps = ka->eps, coef = ka->coef;
        float th[NV], m[NV], vv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const size_t o = (size_t)r * D + lane + 64 * v;
            th[v] = fT[o]; m[v] = fm[o]; vv[v] = fv[o];
        }
        float sq = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const size_t o = (size_t)r * D + lane + 64 * v;
            float g = (s[v] + acc[v]) * scale;
            if (c) { g = fmaf(coef * (float)c, th[v], g); sq = fmaf(th[v], th[v], sq); fdE[o] = 0.f; }
            adam1(th[v], m[v], vv[v], g, lr_t, b1, b2, eps);
            fT[o] = th[v]; fm[o] = m[v]; fv[o] = vv[v];
        }
        if (c) {                                                 // wave-uniform
            sq = wave_sum(sq);
            if (lane == 0) {
                atomicAdd(ka->fz.emb + (r & 2047), (double)c * (double)sq);
                cnt[r] = 0;
            }
        }
        return;
    }
    float *Y = ka->Y, *S_out = ka->S_out;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const size_t o = (size_t)r * D + lane + 64 * v;
        if (Y) Y[o] = acc[v];
        if (S_out) S_out[o] = (s[v] + acc[v]) * scale;
    }
}

template <int D, bool FUSE>
__global__ __launch_bounds__(256) void k_spmm_stream(const StreamArgs A) {
    const int4 *__restrict__ chunk_desc = A.chunk_desc;
    const int32_t *__restrict__ empties = A.empties;
    const int2 *__restrict__ pcw = A.pcw;
    const float *__restrict__ X = A.X;
    const float *S_in = A.S_in;                                  // (the rounds find it behind X: see StreamArgs)
    struct { int n_chunks; float scale; } P = {A.n_chunks, A.scale};
    using G = RowGeom<D>;
    constexpr int NV = G::NV;
    constexpr int NE = 32 / NV;                                  // entries per round of gathers (32 row registers)
    static_assert(!G::kHalf, "d = 32 runs the row kernel");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wid);
    if (w >= P.n_chunks) return;
    const int4 cd = chunk_desc[w];                               // {first sub-batch, end, first empty row | n << 24, slot or -1}
    const int sb0 = cd.x, sb1 = cd.y;
    // The index stream travels through LDS.  (Round 4 measured what the scalar path costs: with the gathers compiled out
    // the kernel took as long as with them -- 27 MB of (index, weight) pairs through the scalar data caches, one 64-byte
    // line per miss and a handful of misses in flight per cache, is ~50 us whatever else happens; with the scalar loads
    // compiled out it took 36.)  A window of 128 entries = 1 KB is ONE LDS-DMA instruction (16 bytes per lane, no
    // registers, asynchronous); three windows per wave form a ring, filled two windows ahead; the rounds read the pairs
    // back as broadcasts (ds_read_b128: two entries per instruction).
    constexpr int kWin = 128, kRing = 3;
    __shared__ int4 ring[4][kRing][kWin / 2];                    // [wave][slot][pair of entries]
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int n_rounds = (sb1 - sb0) * (32 / NE);
    const int n_win = (sb1 - sb0 + 3) / 4;
    auto fill = [&](int j) {                                     // window j of the chunk -> ring slot j % kRing (no wait)
        const char *src = reinterpret_cast<const char *>(pcw) + ((size_t)sb0 * 32 + (size_t)j * kWin) * 8 + lane * 16;
        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)&ring[wid][j % kRing][0], 16, 0, 0);
    };
    if (n_win > 0) fill(0);
    if (n_win > 1) fill(1);
    const float zero[NV] = {};
    // rows without neighbours: A X = 0
    for (int k = cd.z & 0xffffff, k1 = k + ((uint32_t)cd.z >> 24); k < k1; ++k) {
        const int r = empties[k];
        float s[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) s[v] = S_in[(size_t)r * D + lane + 64 * v];
        stream_row_out<D, FUSE>(r, zero, s, lane, P.scale);
    }
    float acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    constexpr int NG = NE / 8;                                   // groups of 8 entries per round: a row can only end at a group's last entry
    constexpr int RPW = kWin / NE;                               // rounds per window
    const uint32_t lane4 = (uint32_t)lane * 4u;
    for (int k = 0; k < n_rounds; ++k) {                         // wave-uniform
        if (k % RPW == 0) {
            // a new window: it was asked for two windows ago.  Every gather issued since has been waited for, so this wait
            // is for the fills alone; then the slot of the window before is free for the window after next.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (k / RPW + 2 < n_win) fill(k / RPW + 2);
        }
        const int4 *win = &ring[wid][(k / RPW) % kRing][(k % RPW) * (NE / 2)];
        float x[NE][NV], wgt[NE];
        int cend[NG];
#pragma unroll
        for (int e = 0; e < NE; e += 2) {
            const int4 q = win[e / 2];                           // broadcast: every lane reads the same two entries
            wgt[e] = __int_as_float(q.y); wgt[e + 1] = __int_as_float(q.w);
            const int c0 = q.x, c1 = q.z;
#pragma unroll
            for (int v = 0; v < NV; ++v) x[e][v] = ld_elem_v<D>(X, c0, lane4 + 256u * v);
            // a group's last entry may be an end marker (bit 31): it gathers S_in, which sits s_in_bytes behind X
            uint32_t extra = 0u;
            if ((e + 1) % 8 == 7) { cend[(e + 1) / 8] = c1; extra = (uint32_t)(c1 >> 31) & A.s_in_bytes; }
#pragma unroll
            for (int v = 0; v < NV; ++v) x[e + 1][v] = ld_elem_v<D>(X, c1, lane4 + 256u * v + extra);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int e = 8 * g; e < 8 * g + 7; ++e) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = fmaf(wgt[e], x[e][v], acc[v]);
            }
            const int c7 = __builtin_amdgcn_readfirstlane(cend[g]);
            if (c7 >= 0) {                                       // wave-uniform: the usual case, an entry like the others
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = fmaf(wgt[8 * g + 7], x[8 * g + 7][v], acc[v]);
            } else if (__builtin_amdgcn_readfirstlane(__float_as_int(wgt[8 * g + 7])) == 0) {   // the row's end marker: x = S_in of the row
                stream_row_out<D, FUSE>(c7 & 0x7fffffff, acc, x[8 * g + 7], lane, P.scale);
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = 0.f;
            }                                                    // (weight bits 1: the end of a piece, always its chunk's last entry)
        }
    }
    const int slot = cd.w;                                       // >= 0: the chunk is a piece of a hub row, this is its slot
    if (slot < 0) return;
    // A piece of a hub row (see k_spmm_row): partial rows travel write-through, whoever arrives last sums them in slot
    // order -- the pieces of a group, then the groups of the row -- and finishes the row.
    const auto *ka = (const __attribute__((address_space(4))) StreamArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const StreamHub hub = {ka->hub.plan, ka->hub.off, 0, ka->hub.arrivals, ka->hub.slab};
    StreamHeader sh;
    {
        const int32_t *hp = hub.plan + hub.off;
        sh.magic = hp[0]; sh.n_chunks = hp[1]; sh.n_sb = hp[2]; sh.n_empty = hp[3]; sh.n_slots = hp[4]; sh.n_groups = hp[5];
        sh.n_split = hp[6]; sh.n_entries = hp[7];
    }
    const StreamView sv = view_stream(hub.plan, hub.off, sh);
    float *slab = hub.slab;
    int32_t *arrivals = hub.arrivals;
    auto publish = [&](int at) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
            __hip_atomic_store(slab + (size_t)at * D + lane + 64 * v, acc[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto last_of = [&](int counter, int expected) {
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(arrivals + counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != expected - 1) return false;
        if (lane == 0) __hip_atomic_store(arrivals + counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    auto sum_range = [&](int p0, int p1) {
        float p[kGroup][NV];
#pragma unroll
        for (int q = 0; q < kGroup; ++q) {
            const int sq = p0 + q < p1 ? p0 + q : p1 - 1;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                p[q][v] = __hip_atomic_load(slab + (size_t)sq * D + lane + 64 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.f;
#pragma unroll
        for (int q = 0; q < kGroup; ++q) {
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += p0 + q < p1 ? p[q][v] : 0.f;
        }
    };
    const int g = sv.slot_group[slot];
    const int s0 = sv.group_slot0[g], s1 = sv.group_slot0[g + 1];
    const int k = sv.group_split[g];
    const int g0 = sv.split_group0[k], g1 = sv.split_group0[k + 1];
    if (s1 - s0 > 1) {
        publish(slot);
        if (!last_of(g, s1 - s0)) return;
        sum_range(s0, s1);
    }
    if (g1 - g0 > 1) {
        publish(sh.n_slots + g);
        if (!last_of(sh.n_groups + k, g1 - g0)) return;
        for (int q0 = g0; q0 < g1; q0 += kGroup) {
            float part[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) part[v] = q0 == g0 ? 0.f : acc[v];
            sum_range(sh.n_slots + q0, sh.n_slots + (q0 + kGroup < g1 ? q0 + kGroup : g1));
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += part[v];
        }
    }
    const int r = sv.split_row[k];
    float s[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = S_in[(size_t)r * D + lane + 64 * v];
    stream_row_out<D, FUSE>(r, acc, s, lane, P.scale);
}

// out = in * scale   (n_layers == 0 degenerate case) -- float4 per lane
__global__ void k_scale_copy(size_t n_vec, const float *__restrict__ in, float *__restrict__ out, float scale) {
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * blockDim.x) {
        const float4 x = ld4(in + v * 4);
        st4(out + v * 4, make_float4(x.x * scale, x.y * scale, x.z * scale, x.w * scale));
    }
}

// what the hub reduction needs behind the three layer buffers: partial rows of pieces and groups + arrival counters, for
// whichever of the plan's two schedules (row items / entry stream) has more of them
static void plan_extras(const void *plan_host, size_t &slab_rows, size_t &counters) {
    const PlanHeader *h = static_cast<const PlanHeader *>(plan_host);
    slab_rows = (size_t)h->n_slots + h->n_groups;
    counters = (size_t)h->n_groups + h->n_split;
    if (h->reserved > 0) {
        const StreamHeader *sh = reinterpret_cast<const StreamHeader *>(static_cast<const int32_t *>(plan_host) + h->reserved);
        slab_rows = std::max(slab_rows, (size_t)sh->n_slots + sh->n_groups);
        counters = std::max(counters, (size_t)sh->n_groups + sh->n_split);
    }
}

// work: [4*N*d] two pairs [layer output | running layer sum] used alternately (no layer updates its input in place), [n_slots*d] slab and [n_split] arrival counters (zero between calls) when a plan is given.
// sp (may be NULL) with mode kSparseOut: only the flagged rows of E are wanted (computed in the last layer);
// with mode kSparseIn: E0 is row-sparse, only its flagged rows are non-zero (and only they are read).
int launch_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col, const float *val,
                     const void *plan_dev, const void *plan_host_header, const float *E0, float *E, float *work,
                     hipStream_t st, const SparseCtx *sp, int sparse_mode, const AdamFuse *fuse) {
    const size_t nd = (size_t)N * d;
    // layer buffers as PAIRS [X | S]: layer l reads (X, S_in) from one pair -- or E0 for both -- and writes (Y, S_out)
    // into the other, so that S_in == X + nd wherever it is not X itself (k_spmm_stream gathers S_in through X)
    float *pairX[2] = {work, work + 2 * nd}, *pairS[2] = {work + nd, work + 3 * nd};
    const float inv = 1.0f / (float)(n_layers + 1);
    if (n_layers == 0) {
        k_scale_copy<<<1024, 256, 0, st>>>(nd / 4, E0, E, 1.0f);
        MACR_CHECK_LAUNCH("scale_copy", st);
        return MACR_OK;
    }
    MACR_REQUIRE(nd * 4 < ((size_t)1 << 32), MACR_E_UNSUPPORTED, "lgcn propagate: table of %zu bytes (rows are addressed with 32-bit byte offsets)", nd * 4);
    PlanHeader ph = {};
    if (plan_dev) ph = *static_cast<const PlanHeader *>(plan_host_header);
    SpmmArgs a = {};
    a.N = N; a.rowptr = rowptr; a.col = col; a.val = val;
    if (plan_dev) {
        const PlanView pv = view_plan(plan_dev, ph);
        a.n_items = ph.n_items; a.n_slots = ph.n_slots;
        a.items = pv.item; a.slot_group = pv.slot_group; a.group_slot0 = pv.group_slot0; a.group_split = pv.group_split;
        a.split_group0 = pv.split_group0; a.n_groups = ph.n_groups;
        a.slab = work + 4 * nd;
        size_t slab_rows, counters;
        plan_extras(plan_host_header, slab_rows, counters);
        a.arrivals = reinterpret_cast<int32_t *>(a.slab + slab_rows * d);
    }
    // the entry stream (k_spmm_stream) runs the dense layers of d >= 64 when the plan carries one
    const bool have_stream = plan_dev && ph.reserved > 0 && d >= 64;
    StreamHeader sh = {};
    StreamView sv = {};
    if (have_stream) {
        sh = *reinterpret_cast<const StreamHeader *>(static_cast<const int32_t *>(plan_host_header) + ph.reserved);
        sv = view_stream(plan_dev, ph.reserved, sh);
    }
    const int n_waves = plan_dev ? ph.n_items : N;
    const float *X = E0;
    const float *S_in = E0;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        float *Y = last ? nullptr : pairX[l & 1];
        float *sum = pairS[l & 1];
        const float scale = last ? inv : 1.0f;        // running sum lives in E; first layer reads E0 as S_in
        // The first backward layer's input (dE) is zero outside the batch's rows: the row-sparse form reads only the flagged
        // rows (a flag lookup per entry, masked gathers), the dense form reads everything and adds zeros -- the same sums up
        // to the sign of a zero.  With the records of the short rows (d = 64) the dense form is the faster one: 38 against
        // 46 us per layer, LightGCN step 199.1 -> 190.3 us at the Yelp2018 shape, same box (profiles/r05_lgcn_bwd1_dense_ab.txt).
        // MACR_LGCN_BWD1_DENSE=0 / 1 forces either (A/B switch); other widths keep the row-sparse form.
        static const char *bwd1_env = getenv("MACR_LGCN_BWD1_DENSE");
        const bool bwd1_dense = bwd1_env && (bwd1_env[0] == '0' || bwd1_env[0] == '1') ? bwd1_env[0] == '1' : (d == 64 && plan_dev != nullptr);
        const int mode = !sp ? kDense
                         : (sparse_mode == kSparseOut && last) ? kSparseOut
                         : (sparse_mode == kSparseIn && l == 0 && !bwd1_dense) ? kSparseIn : kDense;
        const bool fused = last && fuse && fuse->T;          // the optimizer in this layer's epilogue: nothing is stored to E
        a.X = X; a.Y = Y; a.S_in = S_in; a.S_out = fused ? nullptr : last ? E : sum; a.scale = scale;
        if (mode != kDense || fused) a.sp = *sp;
        const AdamFuse af = fused ? *fuse : AdamFuse{};
        const int waves = mode == kSparseOut ? (plan_dev ? ph.n_slots : 0) + 3 * sp->B : n_waves;
        const int grid = (waves + 3) / 4;
        const char *name = mode == kSparseOut ? "spmm_csr_rows" : mode == kSparseIn ? "spmm_csr_sparse" : fused ? "spmm_csr+adam" : "spmm_csr";
        // (the fused layer stays with the row kernel: there the optimizer's operands travel with the row's first gathers,
        // in the stream kernel they would be a dependent trip per row end -- measured 67 against 62 us)
        static const bool stream_fused = getenv("MACR_SPMM_STREAM_FUSED") && atoi(getenv("MACR_SPMM_STREAM_FUSED")) != 0;
        if (have_stream && mode == kDense && (!fused || stream_fused) && sh.magic == kStreamMagic && (a.S_in == a.X || a.S_in == a.X + nd) &&
            2 * nd * 4 < ((size_t)1 << 32)) {
            StreamArgs sa = {};
            sa.chunk_desc = sv.chunk_desc; sa.empties = sv.empties; sa.pcw = sv.pcw;
            sa.s_in_bytes = a.S_in == a.X ? 0u : (uint32_t)(nd * 4);    // (S_in is X itself, or sits N rows behind it: the pair layout)
            sa.X = a.X; sa.S_in = a.S_in; sa.n_chunks = sh.n_chunks; sa.scale = scale;
            sa.Y = a.Y; sa.S_out = a.S_out; sa.b1 = af.b1; sa.b2 = af.b2; sa.eps = af.eps; sa.coef = af.coef;
            sa.hub = StreamHub{static_cast<const int32_t *>(plan_dev), ph.reserved, 0, a.arrivals, a.slab};
            sa.fz = StreamFuse{fused ? sp->cnt : nullptr, af.T, af.m, af.v, af.scal, af.dE, af.emb_acc};
            const int sgrid = (sh.n_chunks + 3) / 4;
#define MACR_STREAM_ARGS sa
#define MACR_STREAM_ROW(D_)                                                                       \
    do {                                                                                          \
        if (fused) k_spmm_stream<D_, true><<<sgrid, 256, 0, st>>>(MACR_STREAM_ARGS);              \
        else k_spmm_stream<D_, false><<<sgrid, 256, 0, st>>>(MACR_STREAM_ARGS);                   \
    } while (0)
            switch (d) {
                case 64: MACR_STREAM_ROW(64); break;
                case 128: MACR_STREAM_ROW(128); break;
                case 256: MACR_STREAM_ROW(256); break;
            }
#undef MACR_STREAM_ROW
#undef MACR_STREAM_ARGS
            MACR_CHECK_LAUNCH(fused ? "spmm_stream+adam" : "spmm_stream", st);
            X = Y;
            S_in = sum;
            continue;
        }
        // dense layers at d = 64: the plan's short rows run from its records, several to a wave (MACR_SPMM_RECORDS=0: one row per wave)
        static const bool rec_on = !(getenv("MACR_SPMM_RECORDS") && atoi(getenv("MACR_SPMM_RECORDS")) == 0);
        const bool rec = rec_on && plan_dev && mode == kDense && d == 64 && ph.rec_off > 0 && ph.n_single > 0;
        const int n_rec = ph.n_rec[0] + ph.n_rec[1] + ph.n_rec[2] + ph.n_rec[3];
        const int rgrid = rec ? (ph.n_single + n_rec + 3) / 4 : grid;
        const int32_t *records = rec ? static_cast<const int32_t *>(plan_dev) + ph.rec_off : nullptr;
        const SpmmScalars ps = {a.N, a.n_items, a.n_slots, a.n_groups, a.scale, a.sp.B, a.sp.n_users, a.sp.chunk, a.sp.count,
                                af.b1, af.b2, af.eps, af.coef, rec ? ph.n_single : 0, ph.n_rec[0], ph.n_rec[1], ph.n_rec[2], ph.n_rec[3]};
#define MACR_SPMM_ARGS ps, a.rowptr, a.col, a.val, a.items, a.slot_group, a.group_slot0, a.group_split, a.split_group0, \
                       a.arrivals, a.X, a.Y, a.S_in, a.S_out, a.slab, a.sp.cnt, a.sp.u, a.sp.i, a.sp.j, af.T, af.m, af.v,  \
                       af.scal, af.dE, af.emb_acc, records
#define MACR_SPMM_ROW(D_)                                                                                         \
    do {                                                                                                          \
        if (mode == kSparseOut) k_spmm_row<D_, kSparseOut, false><<<grid, 256, 0, st>>>(MACR_SPMM_ARGS);          \
        else if (mode == kSparseIn) k_spmm_row<D_, kSparseIn, false><<<grid, 256, 0, st>>>(MACR_SPMM_ARGS);       \
        else if (fused) k_spmm_row<D_, kDense, true><<<rgrid, 256, 0, st>>>(MACR_SPMM_ARGS);                       \
        else k_spmm_row<D_, kDense, false><<<rgrid, 256, 0, st>>>(MACR_SPMM_ARGS);                                 \
    } while (0)
        switch (d) {
            case 32: MACR_SPMM_ROW(32); break;
            case 64: MACR_SPMM_ROW(64); break;
            case 128: MACR_SPMM_ROW(128); break;
            case 256: MACR_SPMM_ROW(256); break;
        }
#undef MACR_SPMM_ROW
#undef MACR_SPMM_ARGS
        MACR_CHECK_LAUNCH(name, st);
        X = Y;
        S_in = sum;
    }
    return MACR_OK;
}

static int plan_chunk() {
    static const int c = getenv("MACR_SPMM_CHUNK") && atoi(getenv("MACR_SPMM_CHUNK")) >= 64 ? atoi(getenv("MACR_SPMM_CHUNK")) : kChunk;
    return c;
}

// Eight ranges of source rows of equal entry mass over the rows longer than `hub` (see build_stream: XCD affinity).
static void source_octants(int N, const int32_t *rowptr, const int32_t *col, int hub, int32_t (&bound)[kNumXcd + 1]) {
    std::vector<int64_t> mass((size_t)N + 1, 0);
    int64_t total = 0;
    for (int r = 0; r < N; ++r)
        if (rowptr[r + 1] - rowptr[r] > hub)
            for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) { ++mass[col[e]]; ++total; }
    bound[0] = 0;
    int64_t run = 0; int x = 1;
    for (int c = 0; c < N && x < kNumXcd; ++c) {
        run += mass[c];
        while (x < kNumXcd && run * kNumXcd >= total * x) bound[x++] = c + 1;
    }
    for (; x <= kNumXcd; ++x) bound[x] = N;
    bound[kNumXcd] = N;
}
static bool plan_octants() {
    static const bool on = !(getenv("MACR_SPMM_OCTANTS") && atoi(getenv("MACR_SPMM_OCTANTS")) == 0);
    return on;
}
// `queues[x]`: what should run on XCD x (block j of a launch lands on XCD j % 8 and takes four consecutive work items);
// `free_items`: what may run anywhere.  Returns the launch order: block j = four items of queue j % 8, the shortest queue
// takes the next free item, an empty queue borrows from the fullest.
static std::vector<int32_t> xcd_order(std::vector<int32_t> (&q)[kNumXcd], const std::vector<int32_t> &free_items) {
    int turn = 0;
    for (int32_t k : free_items) {
        int best = turn;
        for (int x = 0; x < kNumXcd; ++x) if (q[(turn + x) % kNumXcd].size() < q[best].size()) best = (turn + x) % kNumXcd;
        q[best].push_back(k);
        turn = (best + 1) % kNumXcd;
    }
    size_t total = 0, at[kNumXcd] = {};
    for (int x = 0; x < kNumXcd; ++x) total += q[x].size();
    std::vector<int32_t> order;
    order.reserve(total);
    for (int blk = 0; order.size() < total; ++blk) {
        int x = blk % kNumXcd;
        for (int i4 = 0; i4 < 4 && order.size() < total; ++i4) {
            if (at[x] >= q[x].size()) {
                int donor = -1;
                for (int y = 0; y < kNumXcd; ++y)
                    if (at[y] < q[y].size() && (donor < 0 || q[y].size() - at[y] > q[donor].size() - at[donor])) donor = y;
                x = donor;
            }
            order.push_back(q[x][at[x]++]);
        }
    }
    return order;
}

static void build_plan(int N, const int32_t *rowptr, const int32_t *col, const float *val, std::vector<int32_t> &out) {
    struct Item { int32_t row, beg, end, slot; };
    std::vector<Item> items;
    std::vector<int32_t> slot_group, group_slot0, group_split, split_group0, split_row, piece_xcd;
    const int chunk = plan_chunk();
    const bool octants = col && plan_octants();
    int32_t bound[kNumXcd + 1] = {};
    if (octants) source_octants(N, rowptr, col, chunk, bound);
    auto octant_of = [&](int32_t c) { int x = 0; while (x + 1 < kNumXcd && c >= bound[x + 1]) ++x; return x; };
    int n_slots = 0;
    for (int r = 0; r < N; ++r) {
        const int beg = rowptr[r], end = rowptr[r + 1];
        if (end - beg <= chunk) {
            items.push_back({r, beg, end, -1});
        } else {
            split_group0.push_back((int32_t)group_split.size());
            int in_group = kGroup;
            for (int b = beg; b < end;) {
                if (in_group == kGroup) {                       // a new group of this row
                    group_slot0.push_back(n_slots);
                    group_split.push_back((int32_t)split_row.size());
                    in_group = 0;
                }
                slot_group.push_back((int32_t)group_split.size() - 1);
                int pe = b + chunk < end ? b + chunk : end;
                int x = -1;
                if (octants) {                                  // a piece stays inside one range of source rows
                    x = octant_of(col[b]);
                    while (pe > b + 1 && col[pe - 1] >= bound[x + 1]) --pe;
                }
                piece_xcd.push_back(x);
                items.push_back({r, b, pe, n_slots++});
                ++in_group;
                b = pe;
            }
            split_row.push_back(r);
        }
    }
    group_slot0.push_back(n_slots);
    split_group0.push_back((int32_t)group_split.size());
    // (row order instead of length order: 4 % less FETCH_SIZE, same or longer layer time -- measured, not kept)
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
        const bool pa = a.slot >= 0, pb = b.slot >= 0;          // the pieces of the split rows first (in slot order)
        if (pa != pb) return pa;
        if (pa) return false;
        return a.end - a.beg > b.end - b.beg;                   // then the longest rows
    });
    if (octants && n_slots > 0) {
        // the pieces (the first n_slots items, in slot order) in XCD order: piece w of the launch is wave w, block w / 4
        std::vector<int32_t> q[kNumXcd], none;
        for (int k = 0; k < n_slots; ++k) q[piece_xcd[items[k].slot]].push_back(k);
        const std::vector<int32_t> order = xcd_order(q, none);
        std::vector<Item> pieces(items.begin(), items.begin() + n_slots);
        for (int k = 0; k < n_slots; ++k) items[k] = pieces[order[k]];
    }
    PlanHeader h = {};
    h.magic = kPlanMagic; h.n_items = (int32_t)items.size(); h.n_split = (int32_t)split_row.size(); h.n_slots = n_slots; h.N = N;
    h.chunk = chunk; h.n_groups = (int32_t)group_split.size();
    out.assign(reinterpret_cast<int32_t *>(&h), reinterpret_cast<int32_t *>(&h) + sizeof(h) / 4);
    for (const Item &it : items) { out.push_back(it.row); out.push_back(it.beg); out.push_back(it.end); out.push_back(it.slot); }
    out.insert(out.end(), slot_group.begin(), slot_group.end());
    out.insert(out.end(), group_slot0.begin(), group_slot0.end());
    out.insert(out.end(), group_split.begin(), group_split.end());
    out.insert(out.end(), split_group0.begin(), split_group0.end());
    out.insert(out.end(), split_row.begin(), split_row.end());
    // records of the short rows (spmm_records): the items are sorted longest first, so the rows of <= kRecEntries neighbours are
    // the tail of the item list; class c packs 8 >> c rows of at most 4 << c neighbours into one record
    if (col && val) {
        size_t first_short = n_slots;
        while (first_short < items.size() && items[first_short].end - items[first_short].beg > kRecEntries) ++first_short;
        while (out.size() % 4) out.push_back(0);                 // (records are read as 8-byte pairs: keep them 16-byte aligned)
        PlanHeader *hp = reinterpret_cast<PlanHeader *>(out.data());
        hp->n_single = (int32_t)first_short;
        hp->rec_off = (int32_t)out.size();
        size_t k = items.size();                                 // walk from the shortest rows up: class 0 first
        int32_t n_rec[4] = {0, 0, 0, 0};
        for (int c = 0; c < 4; ++c) {
            const int R = 8 >> c, E = kRecEntries / R;
            size_t lo = k;
            while (lo > first_short && items[lo - 1].end - items[lo - 1].beg <= E) --lo;     // rows of this class: [lo, k)
            for (size_t r0 = lo; r0 < k; r0 += R) {
                int32_t recbuf[kRecInts];
                for (int q = 0; q < 16; ++q) recbuf[q] = -1;
                for (int q = 0; q < 2 * kRecEntries; ++q) recbuf[16 + q] = 0;
                for (int q = 0; q < R && r0 + q < k; ++q) {
                    const Item &it = items[r0 + q];
                    recbuf[q] = it.row;
                    for (int e = 0; e < it.end - it.beg; ++e) {
                        recbuf[16 + 2 * (q * E + e)] = col[it.beg + e];
                        memcpy(&recbuf[16 + 2 * (q * E + e) + 1], &val[it.beg + e], 4);
                    }
                }
                out.insert(out.end(), recbuf, recbuf + kRecInts);
                ++n_rec[c];
            }
            k = lo;
        }
        hp = reinterpret_cast<PlanHeader *>(out.data());
        for (int c = 0; c < 4; ++c) hp->n_rec[c] = n_rec[c];
    }
}


// The entry stream of k_spmm_stream (see there).  Appends the stream section (64-byte aligned) to `out`, the row plan.
static int stream_target() {
    static const int t = getenv("MACR_SPMM_T") && atoi(getenv("MACR_SPMM_T")) >= 32 ? atoi(getenv("MACR_SPMM_T")) : 256;
    return t;
}
static int stream_hub() {
    static const int t = getenv("MACR_SPMM_HUB") && atoi(getenv("MACR_SPMM_HUB")) >= 32 ? atoi(getenv("MACR_SPMM_HUB")) : kStreamHubDefault;
    return t;
}
static void build_stream(int N, const int32_t *rowptr, const int32_t *col, const float *val, std::vector<int32_t> &out) {
    const int kStreamHub = stream_hub();
    std::vector<int32_t> pc, prow, chunk_slot, empties, slot_group, group_slot0, group_split, split_group0, split_row;
    std::vector<float> pv;
    const int T = stream_target();
    pc.reserve((size_t)rowptr[N] + 8 * (size_t)N); pv.reserve(pc.capacity());
    // an entry; rows end only at the last entry of a group of 8 (prow: one int per group)
    auto emit = [&](int32_t c, float w) {
        if ((pc.size() & 7) == 0) prow.push_back(-1);
        pc.push_back(c); pv.push_back(w);
    };
    // the end marker of row r (mark >= 0) or of a piece (mark == -2): padded up to a group's last entry.  The padding
    // gathers row `r` with weight 0 (a row the marker reads anyway).
    auto emit_end = [&](int32_t r, int32_t mark) {
        while ((pc.size() & 7) != 7) emit(r, 0.f);
        emit(r, 0.f);
        prow.back() = mark;
    };
    auto pad32 = [&]() { while (pc.size() & 31) emit(0, 0.f); };
    int cost = 0;
    bool open = false;
    std::vector<int32_t> chunk_start;
    auto close_chunk = [&](int slot) {
        if (!open) return;
        pad32();
        chunk_slot.push_back(slot);
        open = false; cost = 0;
    };
    auto open_chunk = [&]() { if (!open) { chunk_start.push_back((int32_t)(pc.size() / 32)); open = true; } };
    int n_slots = 0;
    // XCD affinity of the hub rows' pieces.  A hub row walks its neighbours in ascending order, so a piece covers a RANGE of
    // source rows; block b of a launch runs on XCD b % 8, each with an L2 of its own (4 MB).  The source rows of all hub
    // entries are cut into eight ranges of equal entry mass, a hub row is cut at the range boundaries first, and the
    // chunk order below puts the pieces of range x into blocks that land on XCD x: that L2 then serves one eighth of the
    // source rows (~1 MB of an 8 MB user table) instead of all of them.  Placement is the driver's habit, not a promise:
    // results do not depend on it, only the L2 hit rate does.  MACR_SPMM_OCTANTS=0 keeps stream order.
    const bool octants = plan_octants();
    int32_t bound[kNumXcd + 1];
    source_octants(N, rowptr, col, kStreamHub, bound);
    std::vector<int32_t> chunk_xcd;                               // per chunk: the XCD its block should land on, -1 any
    auto octant_of = [&](int32_t c) { int x = 0; while (x + 1 < kNumXcd && c >= bound[x + 1]) ++x; return x; };
    for (int r = 0; r < N; ++r) {
        const int beg = rowptr[r], end = rowptr[r + 1], deg = end - beg;
        if (deg == 0) { empties.push_back(r); continue; }
        if (deg <= kStreamHub) {
            const int slots = (deg + 1 + 7) / 8 * 8;
            if (open && cost > 0 && cost + slots > T + T / 4) close_chunk(-1);
            open_chunk();
            for (int e = beg; e < end; ++e) emit(col[e], val[e]);
            emit_end(r, r);                                  // gathers S_in[r]
            cost += slots + 4;
            if (cost >= T) close_chunk(-1);
        } else {
            close_chunk(-1);
            split_group0.push_back((int32_t)group_split.size());
            int in_group = kGroup;
            for (int b = beg; b < end;) {
                if (in_group == kGroup) {
                    group_slot0.push_back(n_slots);
                    group_split.push_back((int32_t)split_row.size());
                    in_group = 0;
                }
                slot_group.push_back((int32_t)group_split.size() - 1);
                open_chunk();
                int pe = b + kStreamPiece - 1 < end ? b + kStreamPiece - 1 : end;
                const int x = octant_of(col[b]);
                if (octants)                                      // a piece stays inside one source range
                    while (pe > b + 1 && col[pe - 1] >= bound[x + 1]) --pe;
                for (int e = b; e < pe; ++e) emit(col[e], val[e]);
                emit_end(r, -2);
                close_chunk(n_slots++);
                chunk_xcd.resize(chunk_slot.size(), -1);
                chunk_xcd.back() = octants ? x : -1;
                ++in_group;
                b = pe;
            }
            split_row.push_back(r);
        }
    }
    close_chunk(-1);
    group_slot0.push_back(n_slots);
    split_group0.push_back((int32_t)group_split.size());
    const int n_chunks = (int)chunk_start.size();
    chunk_xcd.resize(n_chunks, -1);
    // chunk_sb[k] = start of chunk k, chunk_sb[n_chunks] = end: chunks are contiguous, so ends = next starts
    std::vector<int32_t> sbv(chunk_start);
    sbv.push_back((int32_t)(pc.size() / 32));
    // the order the chunks are launched in (their descriptors; the stream itself stays in row order): eight queues, one per
    // XCD, the pinned pieces first, the other chunks dealt round; block j takes four chunks of queue j % 8
    std::vector<int32_t> order;
    {
        std::vector<int32_t> q[kNumXcd], free_chunks;
        for (int k = 0; k < n_chunks; ++k) (chunk_xcd[k] >= 0 ? q[chunk_xcd[k]] : free_chunks).push_back(k);
        order = xcd_order(q, free_chunks);
    }
    std::vector<int32_t> chunk_empty(n_chunks + 1);
    for (int k = 0; k <= n_chunks; ++k) chunk_empty[k] = (int32_t)((long long)empties.size() * k / (n_chunks ? n_chunks : 1));
    if (n_chunks == 0) chunk_empty[0] = 0;
    StreamHeader h = {};
    h.magic = kStreamMagic; h.n_chunks = n_chunks; h.n_sb = (int32_t)(pc.size() / 32); h.n_empty = (int32_t)empties.size();
    h.n_slots = n_slots; h.n_groups = (int32_t)group_split.size(); h.n_split = (int32_t)split_row.size();
    h.n_entries = (int32_t)pc.size();
    while ((out.size() * 4) % 64) out.push_back(0);
    reinterpret_cast<PlanHeader *>(out.data())->reserved = (int32_t)out.size();
    out.insert(out.end(), reinterpret_cast<int32_t *>(&h), reinterpret_cast<int32_t *>(&h) + sizeof(h) / 4);
    while (out.size() % 4) out.push_back(0);
    for (int pos = 0; pos < n_chunks; ++pos) {
        const int k = order[pos];
        const int32_t ne = chunk_empty[pos + 1] - chunk_empty[pos];
        out.push_back(sbv[k]); out.push_back(sbv[k + 1]); out.push_back(chunk_empty[pos] | (ne << 24)); out.push_back(chunk_slot[k]);
    }
    out.insert(out.end(), empties.begin(), empties.end());
    out.insert(out.end(), slot_group.begin(), slot_group.end());
    out.insert(out.end(), group_slot0.begin(), group_slot0.end());
    out.insert(out.end(), group_split.begin(), group_split.end());
    out.insert(out.end(), split_group0.begin(), split_group0.end());
    out.insert(out.end(), split_row.begin(), split_row.end());
    while ((out.size() * 4) % 64) out.push_back(0);          // pcw 64-byte aligned RELATIVE to the plan's start (the device copy is allocated aligned)
    for (size_t e = 0; e < pc.size(); ++e) {
        int32_t c = pc[e], wb;
        memcpy(&wb, &pv[e], 4);
        const int32_t mark = (e & 7) == 7 ? prow[e / 8] : -1;
        if (mark != -1) { c |= (int32_t)0x80000000; wb = mark >= 0 ? 0 : 1; }
        out.push_back(c); out.push_back(wb);
    }
    out.insert(out.end(), 256, 0);                           // (a window's fill may read up to 127 entries past the last chunk)
}

}  // namespace macr

using namespace macr;

// ---- plan (host) ----------------------------------------------------------------
static void build_whole_plan(int N, const int32_t *rowptr, const int32_t *col, const float *val, std::vector<int32_t> &v) {
    build_plan(N, rowptr, col, val, v);
    // The entry stream is an option (MACR_SPMM_STREAM=1 at plan time): with the XCD-affine piece order both kernels sit at
    // the same level (stand-alone dense layer at the Yelp2018 shape: row kernel 42.7 us, stream kernel 46.3; LightGCN step
    // 205-209 us either way), and the row kernel needs no second copy of the matrix.
    if (col && val && getenv("MACR_SPMM_STREAM") && atoi(getenv("MACR_SPMM_STREAM")) != 0) build_stream(N, rowptr, col, val, v);
}

extern "C" size_t macr_spmm_plan_bytes(int N, const int32_t *rowptr_host, const int32_t *col_host, const float *val_host) {
    if (N <= 0 || !rowptr_host) return 0;
    std::vector<int32_t> v;
    build_whole_plan(N, rowptr_host, col_host, val_host, v);
    return v.size() * 4;
}

extern "C" int macr_spmm_plan_build(int N, const int32_t *rowptr_host, const int32_t *col_host, const float *val_host,
                                    void *plan_host, size_t plan_bytes) {
    MACR_REQUIRE(N > 0 && rowptr_host && plan_host, MACR_E_INVALID, "spmm_plan_build: bad arguments");
    MACR_REQUIRE((col_host == nullptr) == (val_host == nullptr), MACR_E_INVALID, "spmm_plan_build: col and val go together");
    std::vector<int32_t> v;
    build_whole_plan(N, rowptr_host, col_host, val_host, v);
    MACR_REQUIRE(plan_bytes >= v.size() * 4, MACR_E_WORKSPACE, "spmm_plan_build: buffer %zu < %zu bytes", plan_bytes, v.size() * 4);
    memcpy(plan_host, v.data(), v.size() * 4);
    return MACR_OK;
}

extern "C" size_t macr_lgcn_work_floats(int N, int d, const void *plan_host) {
    size_t n = (size_t)4 * N * d;
    if (plan_host) {
        size_t slab_rows, counters;
        plan_extras(plan_host, slab_rows, counters);
        n += slab_rows * d + counters + 64;
    }
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
                                  macr::as_stream(stream), nullptr, 0, nullptr);
}
