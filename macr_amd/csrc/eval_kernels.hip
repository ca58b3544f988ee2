// Full-catalogue evaluator kernels for gfx950 (MI355X).
//
//   k_score_topk   U.I^T on the fp32 MFMA (v_mfma_f32_32x32x2_f32) with the
//                  (y - c)*sig_i*sig_u epilogue, train-item masking and a running
//                  per-user top-K -- the (U,N) score matrix never exists.
//   k_topk_scores  top-K of a materialised score matrix (drop-in for the reference's
//                  c_top_k_array_index, tools.h:24).
//   k_topk_merge   merge of per-split / per-GPU-shard top-K lists.
//   k_metrics_*    ranking metrics.
//
// Ranking rule everywhere: score descending, exact ties by ascending item id
// (what heapq.nlargest over the ascending candidate list does, macr_mf/train.py:89-104;
// std::partial_sort_copy, tools.h:13-22, leaves ties unspecified).  Implemented by
// sorting 64-bit keys (orderable(score) << 32 | ~id).
#include "common.hpp"

#include <type_traits>

namespace macr {

#ifdef MACR_ABL_COUNT
__device__ unsigned long long g_dbg[8];      // [0] tiles with any candidate (per wave), [1] appended keys, [2] compactions, [3] tiles
#define MACR_DBG_ADD(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_dbg[k], (unsigned long long)(v)); } while (0)
#else
#define MACR_DBG_ADD(k, v) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCap = 64;          // candidate buffer entries per user (= one wave-wide sort)
constexpr int kTileItems = 32;    // MFMA tile: 32 items x 32 users
constexpr int kUnitK = 64;        // k-extent staged in LDS at a time
constexpr int kUnitStride = kUnitK + 1;   // odd stride: conflict-free fragment reads
constexpr int kWavesPerBlock = 8;
constexpr int kUsersPerBlock = 32 * kWavesPerBlock;

// Wave-cooperative compaction of one user's candidate buffer (<= 64 keys, one per lane): keep the
// best K, publish the new count and the admission threshold (K-th best score, or -inf while fewer
// than K candidates exist).  All 64 lanes must call it.
//
// Selection, not sorting: the K-th largest key is found by a most-significant-bit-first radix
// select whose state is a 64-bit lane mask in SGPRs (one v_cmp + a few s_* per bit, no cross-lane
// data movement, early exit as soon as one candidate is left -- usually after 10-20 bits); the
// survivors are then packed to the front with a prefix popcount.  (A 64-lane bitonic sort through
// ds_bpermute cost ~5k cycles per call and dominated the kernel.)  The buffer stays unsorted; only
// the final result is sorted.
__device__ __forceinline__ void compact_buffer(uint64_t *keys, uint32_t *cnt, float *thr, int K,
                                               uint64_t *kth_out = nullptr, uint32_t *shared_thr = nullptr) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t c = *cnt;
    const bool valid = lane < (int)c;
    const uint64_t key = valid ? keys[lane] : 0ull;
    uint64_t kth = 0ull;                                 // 0 = fewer than K candidates: no threshold yet
    if (c >= (uint32_t)K) {                              // (uniform)
        uint64_t cand = __ballot(valid);                 // lanes that may still be the K-th largest
        int remaining = K;
        for (int bit = 63; bit >= 0; --bit) {
            if (__popcll(cand) == 1) break;              // keys are distinct (ids differ): one candidate left
            const uint64_t ones = __ballot((key >> bit) & 1ull) & cand;
            const int n1 = __popcll(ones);
            if (n1 >= remaining) cand = ones;            // the K-th largest has this bit set
            else { remaining -= n1; cand &= ~ones; }
        }
        const int kth_lane = __ffsll((long long)cand) - 1;
        const uint32_t kth_hi = __shfl((uint32_t)(key >> 32), kth_lane, kWave);
        const uint32_t kth_lo = __shfl((uint32_t)key, kth_lane, kWave);
        kth = ((uint64_t)kth_hi << 32) | kth_lo;
        if (c > (uint32_t)K) {
            const bool keep = valid && key >= kth;
            const uint64_t kmask = __ballot(keep);
            // every lane holds its key in a register: overwrite the front of the buffer with the survivors
            if (keep) keys[__popcll(kmask & ((1ull << lane) - 1ull))] = key;
        }
    }
    if (lane == 0) {
        *cnt = c < (uint32_t)K ? c : (uint32_t)K;
        *thr = kth ? orderable_f32((uint32_t)(kth >> 32)) : -INFINITY;
        if (kth_out) *kth_out = kth;
        // publish a lower bound of the GLOBAL K-th best score to the other item splits of this user
        if (shared_thr && kth) atomicMax(shared_thr, (uint32_t)(kth >> 32));
    }
    // other lanes read cnt/thr/keys next: the compiler must not forward values it loaded before
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ----------------------------------------------------------------------------
// k_score_topk
// Replaces sess.run(model.batch_ratings | model.rubi_ratings_both, ...) +
// candidate filtering + ranking (macr_mf/train.py:224-251,:119-138,:89-104;
// macr_lightgcn/utility/batch_test.py:50-93,:124-134; tools.h:13-33).
//
// Grid: blockIdx.x = user_block * n_splits + split.  A block (8 waves) owns 256 query
// users and one contiguous item range; consecutive block ids share the user block and
// differ in split, so with n_splits a multiple of 8 every XCD (block id mod 8) streams
// only its own slice of the item table through its private L2.
// Wave w owns users [32w, 32w+32): their embeddings sit in registers as the MFMA B
// operand (D/2 VGPRs) for the whole kernel.  Items stream through LDS in units of
// 32 items x 64 k (double buffered, one barrier per unit); each unit costs 32
// v_mfma_f32_32x32x2_f32 per wave.  MFMA result layout puts user (lane&31) in both lanes
// l and l+32, each holding 16 of the tile's 32 item scores, so thresholding is 16
// compares per lane with no cross-lane traffic.  Scores above the user's running
// threshold are appended to a 64-entry LDS buffer; when it could overflow the wave
// sorts it (bitonic over 64 lanes) and keeps the best K.
// ----------------------------------------------------------------------------
template <int D, int KIND>
__global__ __launch_bounds__(512, 2) void k_score_topk(
    int U, int n_local, const float *__restrict__ users_tab, const int32_t *__restrict__ user_ids,
    const float *__restrict__ items, const float *__restrict__ sig_u, const float *__restrict__ sig_i, float c,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_idx, int item_offset, int K,
    int n_splits, float *__restrict__ out_val, int32_t *__restrict__ out_idx, uint32_t *shared_thr) {
    constexpr int NKH = D / kUnitK > 0 ? D / kUnitK : 1;     // k-halves per tile (D=32 -> 1 short unit)
    constexpr int UK = D < kUnitK ? D : kUnitK;              // k extent of one unit
    constexpr int NT = UK / 2;                               // MFMA steps per unit
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *s_keys = reinterpret_cast<uint64_t *>(smem);                                 // [256][64]
    float *s_unit = reinterpret_cast<float *>(smem + (size_t)kUsersPerBlock * kCap * 8);   // [2][32][65]
    float *s_sig = s_unit + 2 * kTileItems * kUnitStride;                                  // [3][32]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_sig + 3 * kTileItems);                // [256]
    float *s_thr = reinterpret_cast<float *>(s_cnt + kUsersPerBlock);                      // [256]
    uint64_t *s_kth = reinterpret_cast<uint64_t *>(s_thr + kUsersPerBlock);                // [256]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int ub = blockIdx.x / n_splits, split = blockIdx.x % n_splits;
    const int uslot = wid * 32 + col;                    // user slot inside the block
    const int q = ub * kUsersPerBlock + uslot;           // query index
    const bool q_ok = q < U;

    // item range of this split (multiples of 32 except the last)
    const int tiles_total = (n_local + kTileItems - 1) / kTileItems;
    const int tiles_per_split = (tiles_total + n_splits - 1) / n_splits;
    const int it_lo = min(split * tiles_per_split * kTileItems, n_local);
    const int it_hi = min(it_lo + tiles_per_split * kTileItems, n_local);
    const int n_tiles = (it_hi - it_lo + kTileItems - 1) / kTileItems;

    if (tid < kUsersPerBlock) { s_cnt[tid] = 0; s_thr[tid] = -INFINITY; s_kth[tid] = 0ull; }

    // B operand: lane (col,h) holds user[col][2t+h] for every MFMA step t
    float bfrag[D / 2];
    {
        const float *urow = users_tab + (size_t)(q_ok ? (user_ids ? user_ids[q] : q) : 0) * D;
#pragma unroll
        for (int t = 0; t < D / 2; ++t) bfrag[t] = q_ok ? urow[2 * t + h] : 0.f;
    }
    const float su = (KIND == MACR_SCORE_RUBI_BOTH && q_ok) ? sig_u[q] : 1.0f;

    // Stream order.  Item ids often correlate with popularity (ids are handed out by first appearance), and
    // a monotone score trend along the stream is the worst case of a running top-K (every item beats the
    // threshold).  So the tiles are visited in two ascending passes: every 8th tile first (a sample that
    // spans the whole range and sets a good threshold), then the rest.
    constexpr int kStride = 8;
    const int n_pass_a = (n_tiles + kStride - 1) / kStride;
    auto tile_at = [&](int step) -> int {
        if (step < n_pass_a) return step * kStride;
        const int r = step - n_pass_a;
        return r + r / (kStride - 1) + 1;
    };

    // train-item mask cursor: next masked GLOBAL id >= the current position; reset at the start of pass B
    int mbeg = 0, mpos = 0, mend = 0, mnext = INT_MAX;
    if (mask_ptr && q_ok) {
        mpos = mask_ptr[q]; mend = mask_ptr[q + 1];
        const int lo_gid = it_lo + item_offset;
        int lo = mpos, hi = mend;                  // first entry >= lo_gid
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mask_idx[mid] < lo_gid) lo = mid + 1; else hi = mid; }
        mbeg = mpos = lo;
        mnext = mpos < mend ? mask_idx[mpos] : INT_MAX;
    }
    // admission threshold = the user's current K-th best KEY (score, id); 0 = none yet.  Padding users never admit.
    float thr = q_ok ? -INFINITY : INFINITY;
    int thr_id = -1;                               // with score == thr, only ids below thr_id rank higher
    float gthr = -INFINITY;                        // lower bound of the global K-th best score, from the other splits
#ifdef MACR_ABL_NOADMIT
    thr = INFINITY;
#endif

    // staging: 512 threads x one float4 = 32 items x 64 k
    const int st_row = tid >> 4, st_c4 = tid & 15;
    auto load_unit = [&](int tile, int kh) -> float4 {
        const int it = it_lo + tile * kTileItems + st_row;
        if (it < it_hi && st_c4 * 4 < UK) return ld4(items + (size_t)it * D + kh * kUnitK + st_c4 * 4);
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_unit = [&](int buf, float4 v) {
        float *p = s_unit + (size_t)buf * kTileItems * kUnitStride + st_row * kUnitStride + st_c4 * 4;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    };
    auto load_sig = [&](int tile) -> float {
        const int it = it_lo + tile * kTileItems + tid;
        return (KIND == MACR_SCORE_RUBI_BOTH && tid < kTileItems && it < it_hi) ? sig_i[it] : 0.f;
    };

    const int n_units = n_tiles * NKH;
    if (n_units > 0) {
        const float4 v0 = load_unit(tile_at(0), 0);
        const float sg0 = load_sig(tile_at(0));
        store_unit(0, v0);
        if (tid < kTileItems) s_sig[tid] = sg0;
    }
    __syncthreads();

    // Software pipeline: while the matrix pipe multiplies tile `step`, the VALU scores, masks and
    // thresholds tile `step-1` from the other accumulator -- the epilogue instructions sit between the
    // MFMAs of the same wave (an fp32 32x32x2 MFMA occupies the pipe for 64 cycles; ~8 VALU fit per gap).
    // Only the rare admission path (LDS atomics, compaction) runs outside the MFMA stream.
    struct Prev { int tile; int step; };
    const float kNone = __builtin_nanf("");                // masked / out of range: every compare is false

    auto stage = [&](auto have_cur_t, auto have_prev_t, f32x16 &acc_cur, const f32x16 &acc_prev, int step, Prev prev) {
        constexpr bool have_cur = decltype(have_cur_t)::value;     // compile-time: keeps the MFMA stream branch-free
        constexpr bool have_prev = decltype(have_prev_t)::value;
        const int tile = have_cur ? tile_at(step) : 0;
        const int tile_next = step + 1 < n_tiles ? tile_at(step + 1) : 0;
        // ---- previous tile: mask bits and thresholds first (they feed the interleaved scoring)
        const int it0 = it_lo + prev.tile * kTileItems;            // local id of the previous tile's row 0
        const int gid0 = it0 + item_offset;
        uint32_t tmask = 0;
        if (have_prev) {
            if (prev.step == n_pass_a) {                        // pass B starts again from the front of the range
                mpos = mbeg;
                mnext = mpos < mend ? mask_idx[mpos] : INT_MAX;
            }
#ifndef MACR_ABL_NOMASK
            if (__any(mnext < gid0 + kTileItems)) {             // the cursor also skips tiles this pass jumps over
                while (mnext < gid0 + kTileItems) {
                    if (mnext >= gid0) tmask |= 1u << (mnext - gid0);
                    ++mpos;
                    mnext = mpos < mend ? mask_idx[mpos] : INT_MAX;
                }
            }
#endif
            // other splits' thresholds (relaxed agent-scope load: served by L2, stale values only prune less)
            if (shared_thr && q_ok && (prev.step & 3) == 0) {
                const uint32_t g_bits = __hip_atomic_load(&shared_thr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g_bits) gthr = fmaxf(gthr, orderable_f32(g_bits));
            }
        }
        const int sbuf = (prev.step >= 0 ? prev.step : 0) % 3;
        float s[16];
        uint32_t cand = 0;
        auto score_one = [&](int r) {                            // branch-free: interleaves with the MFMAs
            const int il = (r & 3) + 8 * (r >> 2) + 4 * h;       // item row inside the tile
            float v = acc_prev[r];
            if (KIND == MACR_SCORE_RUBI_BOTH) { v = v - c; v = v * s_sig[sbuf * kTileItems + il]; v = v * su; }
            const bool live = (it0 + il < it_hi) & (((tmask >> il) & 1u) == 0u);
            v = live ? v : kNone;
            s[r] = v;
            const bool in = ((v > thr) | ((v == thr) & (gid0 + il < thr_id))) & (v >= gthr);   // no short-circuit: no branches
            cand |= in ? (1u << r) : 0u;
        };
#pragma unroll
        for (int kh = 0; kh < NKH; ++kh) {            // static kh: bfrag[] stays in registers
            const int unit = step * NKH + kh, buf = unit & 1;
            // prefetch the next unit into registers while this one is multiplied
            const bool has_next = unit + 1 < n_units;
            const int ntile = (kh + 1 < NKH) ? tile : tile_next;
            const int nkh = (kh + 1 < NKH) ? kh + 1 : 0;
            float4 vnext = make_float4(0.f, 0.f, 0.f, 0.f);
            float sgnext = 0.f;
#ifndef MACR_ABL_NOSTAGE
            if (has_next) { vnext = load_unit(ntile, nkh); if (nkh == 0) sgnext = load_sig(ntile); }
#endif

            if (kh == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_cur[r] = 0.f;
            }
            const float *ua = s_unit + (size_t)buf * kTileItems * kUnitStride + col * kUnitStride + h;
#ifndef MACR_ABL_NOMFMA
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (have_cur) acc_cur = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[2 * t], bfrag[kh * NT + t], acc_cur, 0, 0, 0);
                if (kh == 0 && (t * 16) % NT == 0 && have_prev) score_one(t * 16 / NT);
            }
#else
            if (have_cur) acc_cur[0] += ua[0] * bfrag[kh * NT];
            if (kh == 0 && have_prev) {
#pragma unroll
                for (int r = 0; r < 16; ++r) score_one(r);
            }
#endif
            if (kh == 0 && have_prev) {
                // admission, in two half-tiles of 8 registers so that a user gains at most 16 entries per
                // round: the buffer may then fill to kCap-16 = 48 before it has to be compacted.
                // Exact rule: key(s,id) > key(thr,thr_id); and s >= the other splits' bound (ties pass).
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const uint32_t cand_h = cand & (0xffu << (8 * half));
                    const uint32_t n_l = __popc(cand_h);
                    MACR_DBG_ADD(3, 1);
                    if (__any(n_l != 0)) {
                        MACR_DBG_ADD(0, 1);
#ifdef MACR_ABL_COUNT
                        { unsigned tot = n_l; for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m, kWave); MACR_DBG_ADD(1, tot); }
#endif
                        uint32_t end = 0;
                        if (n_l) {
                            uint32_t pos = atomicAdd(&s_cnt[uslot], n_l);      // ONE LDS atomic per lane and round
                            uint64_t *dst = s_keys + (size_t)uslot * kCap;
#pragma unroll
                            for (int r = 8 * half; r < 8 * half + 8; ++r) {
                                if ((cand_h >> r) & 1u) {
                                    const int il = (r & 3) + 8 * (r >> 2) + 4 * h;
                                    dst[pos++] = make_key(s[r], gid0 + il);
                                }
                            }
                            end = pos;              // the later of the user's two lanes sees the full count
                        }
                        const uint64_t bal = __ballot(end > (uint32_t)(kCap - 16));
                        uint32_t todo = (uint32_t)bal | (uint32_t)(bal >> 32);      // users (cols) to compact
                        if (todo) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            const uint32_t mine = (todo >> col) & 1u;
                            while (todo) {
                                const int ucol = __ffs((int)todo) - 1;
                                todo &= todo - 1;
                                const int us = wid * 32 + ucol;
                                const int uq = ub * kUsersPerBlock + us;
                                MACR_DBG_ADD(2, 1);
                                compact_buffer(s_keys + (size_t)us * kCap, &s_cnt[us], &s_thr[us], K, &s_kth[us],
                                               (shared_thr && uq < U) ? shared_thr + uq : nullptr);
                            }
                            if (mine && q_ok) {
                                const uint64_t kk = s_kth[uslot];
                                if (kk) { thr = key_score(kk); thr_id = key_id(kk); }
                            }
                        }
                    }
                }
            }
#ifndef MACR_ABL_NOSTAGE
            if (has_next) {
                store_unit(buf ^ 1, vnext);
                if (nkh == 0 && tid < kTileItems) s_sig[((step + 1) % 3) * kTileItems + tid] = sgnext;
            }
#endif
#ifndef MACR_ABL_NOBARRIER
            __syncthreads();
#endif
        }
        return Prev{tile, have_cur ? step : -1};
    };

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    Prev prev{0, -1};
    using T = std::true_type;
    using F = std::false_type;
    // n_tiles + 1 stages: fill, steady state (two per trip so the accumulators swap statically), drain
    if (n_tiles > 0) {
        prev = stage(T{}, F{}, accA, accB, 0, prev);
        int step = 1;
        for (; step + 1 < n_tiles; step += 2) {
            prev = stage(T{}, T{}, accB, accA, step, prev);
            prev = stage(T{}, T{}, accA, accB, step + 1, prev);
        }
        if (step < n_tiles) {                 // one steady stage left, then drain from B
            prev = stage(T{}, T{}, accB, accA, step, prev);
            prev = stage(F{}, T{}, accA, accB, step + 1, prev);
        } else {                              // drain from A
            prev = stage(F{}, T{}, accB, accA, step, prev);
        }
    }

    // final: sort every user's buffer, write K (score,id) pairs (descending), pad with (-inf,-1)
    for (int ucol = 0; ucol < 32; ++ucol) {
        const int us = wid * 32 + ucol;
        const int qq = ub * kUsersPerBlock + us;
        if (qq >= U) break;
        const uint32_t cc = s_cnt[us];
        uint64_t key = (lane < (int)cc) ? s_keys[(size_t)us * kCap + lane] : 0ull;
        key = wave_sort_desc(key);
        if (lane < K) {
            const size_t o = ((size_t)split * U + qq) * K + lane;
            out_val[o] = key ? key_score(key) : -INFINITY;
            out_idx[o] = key ? key_id(key) : -1;
        }
    }
}

inline size_t score_topk_smem_bytes() {
    return (size_t)kUsersPerBlock * kCap * 8 + 2 * kTileItems * kUnitStride * 4 + 3 * kTileItems * 4 +
           kUsersPerBlock * 4 + kUsersPerBlock * 4 + kUsersPerBlock * 8;
}

// ----------------------------------------------------------------------------
// k_score_matrix: the literal (U,N) score matrix (model.batch_ratings /
// model.rubi_ratings_both), same arithmetic as k_score_topk.  One wave = 32 users x 32 items.
// ----------------------------------------------------------------------------
template <int D, int KIND>
__global__ __launch_bounds__(256) void k_score_matrix(int U, int n_local, const float *__restrict__ users_tab,
                                                      const int32_t *__restrict__ user_ids,
                                                      const float *__restrict__ items,
                                                      const float *__restrict__ sig_u,
                                                      const float *__restrict__ sig_i, float c,
                                                      float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int q = blockIdx.y * 32 + col;
    const int it0 = (blockIdx.x * 4 + wid) * 32;
    if (it0 >= n_local) return;
    const bool q_ok = q < U;
    const float *urow = users_tab + (size_t)(q_ok ? (user_ids ? user_ids[q] : q) : 0) * D;
    const int it = it0 + col;
    const float *irow = items + (size_t)(it < n_local ? it : 0) * D;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int t = 0; t < D / 2; ++t) {
        const float a = it < n_local ? irow[2 * t + h] : 0.f;
        const float b = q_ok ? urow[2 * t + h] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (!q_ok) return;
    const float su = KIND == MACR_SCORE_RUBI_BOTH ? sig_u[q] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int id = it0 + il;
        if (id < n_local) {
            float v = acc[r];
            if (KIND == MACR_SCORE_RUBI_BOTH) { v = v - c; v = v * sig_i[id]; v = v * su; }
            out[(size_t)q * n_local + id] = v;
        }
    }
}

// ----------------------------------------------------------------------------
// k_topk_scores: one wave per row of a materialised (rows, cols) score matrix.
// Replaces c_top_k_array_index (tools.h:24).  HBM-bound: reads rows*cols*4 bytes once.
// Every column is a candidate (also -inf ones, batch_test.py:129), admission by key.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_topk_scores(const float *__restrict__ scores, int cols, int rows, int K,
                                                     int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    __shared__ uint64_t s_keys[4][kCap];
    __shared__ uint64_t s_kth[4];
    __shared__ uint32_t s_cnt[4];
    __shared__ float s_thr[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    if (row >= rows) return;
    const float *src = scores + (size_t)row * cols;
    uint64_t *keys = s_keys[wid];
    if (lane == 0) { s_cnt[wid] = 0; s_thr[wid] = -INFINITY; }
    uint32_t cnt = 0;
    uint64_t thr_key = 0;                           // admission: key > thr_key
    for (int base = 0; base < cols; base += kWave) {
        const int cidx = base + lane;
        const float v = cidx < cols ? src[cidx] : 0.f;
        const uint64_t key = cidx < cols ? make_key(v, cidx) : 0ull;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bool mine = (lane >> 5) == half;
            bool cand = mine && key > thr_key;
            uint64_t bal = __ballot(cand);
            if (bal == 0) continue;
            if (cnt + (uint32_t)__popcll(bal) > (uint32_t)kCap) {
                if (lane == 0) s_cnt[wid] = cnt;
                compact_buffer(keys, &s_cnt[wid], &s_thr[wid], K, &s_kth[wid]);
                cnt = s_cnt[wid];
                thr_key = s_kth[wid];
                cand = mine && key > thr_key;
                bal = __ballot(cand);
            }
            if (cand) {
                const uint32_t pos = cnt + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                keys[pos] = key;
            }
            cnt += (uint32_t)__popcll(bal);
        }
    }
    uint64_t key = (lane < (int)cnt) ? keys[lane] : 0ull;
    key = wave_sort_desc(key);
    if (lane < K) {
        out_idx[(size_t)row * K + lane] = key ? key_id(key) : -1;
        if (out_val) out_val[(size_t)row * K + lane] = key ? key_score(key) : -INFINITY;
    }
}

// ----------------------------------------------------------------------------
// k_topk_merge: one wave per query merges W sorted lists of K into one.
// Also the merge after the RCCL all-gather of per-shard top-K.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_topk_merge(int W, int U, int K, const float *__restrict__ vals,
                                                    const int32_t *__restrict__ idxs,
                                                    const int32_t *__restrict__ fill_ptr,
                                                    const int32_t *__restrict__ fill_idx,
                                                    float *__restrict__ out_val, int32_t *__restrict__ out_idx,
                                                    int32_t *__restrict__ out_cnt) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wid;
    if (q >= U) return;
    const int per_round = (kWave - K) / K;          // lists merged per sort (>= 1 since K <= 32)
    uint64_t best = 0ull;                           // lanes [0,K): current best keys (sorted)
    for (int s0 = 0; s0 < W; s0 += per_round) {
        uint64_t key = lane < K ? best : 0ull;
        const int rel = lane - K;
        if (rel >= 0 && rel < per_round * K) {
            const int s = s0 + rel / K, k = rel % K;
            if (s < W) {
                const size_t o = ((size_t)s * U + q) * K + k;
                const int32_t id = idxs[o];
                if (id >= 0) key = make_key(vals[o], id);
            }
        }
        best = wave_sort_desc(key);
    }
    const uint64_t bal = __ballot(lane < K && best != 0ull);
    int cnt = __popcll(bal);
    if (lane < K) {
        out_val[(size_t)q * K + lane] = best ? key_score(best) : -INFINITY;
        out_idx[(size_t)q * K + lane] = best ? key_id(best) : -1;
    }
    if (lane == 0) {
        if (out_cnt) out_cnt[q] = cnt;
        if (fill_ptr && cnt < K)       // complete with the masked ids, ascending (score -inf)
            for (int e = fill_ptr[q]; e < fill_ptr[q + 1] && cnt < K; ++e, ++cnt) {
                out_idx[(size_t)q * K + cnt] = fill_idx[e];
                out_val[(size_t)q * K + cnt] = -INFINITY;
            }
    }
}

// ----------------------------------------------------------------------------
// branch sigmoid: out[r] = sigmoid(rows[idx?idx[r]:r] . w)      macr_mf/model.py:194-196,:199
// ----------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void k_branch_sigmoid(const float *__restrict__ rows,
                                                        const int32_t *__restrict__ idx, int n,
                                                        const float *__restrict__ w, float *__restrict__ out) {
    constexpr int d = 4 * LPR;
    const int sub = threadIdx.x % LPR;
    const int r = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (r >= n) return;
    const size_t src = idx ? (size_t)idx[r] : (size_t)r;
    const float s = group_sum<LPR>(dot4(ld4(rows + src * d + 4 * sub), ld4(w + 4 * sub)));
    if (sub == 0) out[r] = sigmoid_acc(s);
}

// ----------------------------------------------------------------------------
// metrics
// ----------------------------------------------------------------------------
__device__ __forceinline__ bool in_sorted(const int32_t *a, int n, int32_t x) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == x;
}

// evaluate_foldout.h:16-195 -- [precision|recall|ap|ndcg|mrr] x K prefixes, float accumulators with
// the double sub-expressions of the C++ (1.0*hits/(i+1), 1.0/log2(i+2)).  Thread per query.
__global__ void k_metrics_foldout(int U, int K, const int32_t *__restrict__ rankings,
                                  const int32_t *__restrict__ gt_ptr, const int32_t *__restrict__ gt_idx,
                                  float *__restrict__ results, int hr_in_ap_slot) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int32_t *rank = rankings + (size_t)u * K;
    const int32_t *truth = gt_idx + gt_ptr[u];
    const int truth_len = gt_ptr[u + 1] - gt_ptr[u];
    float *res = results + (size_t)u * 5 * K;
    int hits = 0; float sum_pre = 0.f, dcg = 0.f, idcg = 0.f, rr = 0.f; bool found = false;
    for (int i = 0; i < K; ++i) {
        const bool hit = rank[i] >= 0 && in_sorted(truth, truth_len, rank[i]);
        if (hit) {
            hits += 1;
            const float pre = (float)(1.0 * hits / (i + 1));
            sum_pre += pre;
            dcg = (float)((double)dcg + 1.0 / log2((double)(i + 2)));
            if (!found) { rr = (float)(1.0 / (i + 1)); found = true; }
        }
        if (i < truth_len) idcg = (float)((double)idcg + 1.0 / log2((double)(i + 2)));
        res[0 * K + i] = (float)(1.0 * hits / (i + 1));
        res[1 * K + i] = (float)(1.0 * hits / truth_len);
        res[2 * K + i] = sum_pre / (float)truth_len;
        // batch_test.py:143-149 overwrites the AP slot with HR := 1[recall@k != 0]
        if (hr_in_ap_slot) res[2 * K + i] = (res[1 * K + i] != 0.f) ? 1.0f : 0.0f;
        res[3 * K + i] = dcg / idcg;
        res[4 * K + i] = rr;
    }
}

// macr_mf/train.py:32-117 in float64: per query {precision, recall, ndcg, hit} x Ks.
struct KsArg { int32_t k[8]; int n; };
__global__ void k_metrics_mf(int U, int Kmax, const int32_t *__restrict__ rankings,
                             const int32_t *__restrict__ cnt, const int32_t *__restrict__ gt_ptr,
                             const int32_t *__restrict__ gt_idx, KsArg Ks, double *__restrict__ out) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int32_t *rank = rankings + (size_t)u * Kmax;
    const int32_t *truth = gt_idx + gt_ptr[u];
    const int truth_len = gt_ptr[u + 1] - gt_ptr[u];
    const int len = cnt ? cnt[u] : Kmax;
    for (int qk = 0; qk < Ks.n; ++qk) {
        const int K = Ks.k[qk];
        const int m = len < K ? len : K;
        double hits = 0, dcg = 0, dcg_max = 0;
        for (int i = 0; i < m; ++i)
            if (rank[i] >= 0 && in_sorted(truth, truth_len, rank[i])) { hits += 1.0; dcg += 1.0 / log2((double)(i + 2)); }
        const int lim = truth_len < K ? truth_len : K;
        for (int i = 0; i < lim; ++i) dcg_max += 1.0 / log2((double)(i + 2));
        double *o = out + ((size_t)u * 4) * Ks.n;
        o[0 * Ks.n + qk] = m > 0 ? hits / m : NAN;
        o[1 * Ks.n + qk] = hits / truth_len;
        o[2 * Ks.n + qk] = dcg_max != 0 ? dcg / dcg_max : 0.0;
        o[3 * Ks.n + qk] = hits > 0 ? 1.0 : 0.0;
    }
}

// column means in float64, one block per column, fixed-shape tree => deterministic
template <typename T>
__global__ __launch_bounds__(256) void k_colmean(const T *__restrict__ in, int rows, int cols, double *__restrict__ out) {
    __shared__ double red[4];
    const int cidx = blockIdx.x;
    double s = 0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)in[(size_t)r * cols + cidx];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[cidx] = (red[0] + red[1] + red[2] + red[3]) / (double)rows;
}

}  // namespace macr

// ============================================================================
// C ABI
// ============================================================================
using namespace macr;

#ifdef MACR_ABL_COUNT
extern "C" void macr_dbg_counters(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(macr::g_dbg), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(macr::g_dbg), z, sizeof(z));
}
#endif

extern "C" size_t macr_score_topk_workspace_bytes(int U) { return U > 0 ? (size_t)U * 4 : 0; }

extern "C" int macr_score_topk_splits(int U, int n_local, int d) {
    (void)d;
    if (U <= 0 || n_local <= 0) return 1;
    const int ublocks = (U + kUsersPerBlock - 1) / kUsersPerBlock;
    const int tiles = (n_local + kTileItems - 1) / kTileItems;
    // One block (8 waves, 145 KB LDS) per CU, 256 CUs.  Cost model in units of item tiles per block:
    // rounds * (tiles/s + warm-up), where every split re-pays the running top-K warm-up
    // (~K ln(n/K) extra admissions, worth about a dozen tiles).  Pick the cheapest s.
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= 64 && s <= tiles; ++s) {
        const long long blocks = (long long)ublocks * s;
        const double rounds = (double)((blocks + 255) / 256);
        const double cost = rounds * ((double)((tiles + s - 1) / s) + 12.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

#define MACR_DISPATCH_DK(d, kind, ...)                                                              \
    switch (d) {                                                                                    \
        case 32:  if (kind == MACR_SCORE_NORMAL) { constexpr int D = 32,  KIND = 0; __VA_ARGS__; }  \
                  else { constexpr int D = 32,  KIND = 1; __VA_ARGS__; } break;                     \
        case 64:  if (kind == MACR_SCORE_NORMAL) { constexpr int D = 64,  KIND = 0; __VA_ARGS__; }  \
                  else { constexpr int D = 64,  KIND = 1; __VA_ARGS__; } break;                     \
        case 128: if (kind == MACR_SCORE_NORMAL) { constexpr int D = 128, KIND = 0; __VA_ARGS__; }  \
                  else { constexpr int D = 128, KIND = 1; __VA_ARGS__; } break;                     \
        case 256: if (kind == MACR_SCORE_NORMAL) { constexpr int D = 256, KIND = 0; __VA_ARGS__; }  \
                  else { constexpr int D = 256, KIND = 1; __VA_ARGS__; } break;                     \
    }

extern "C" int macr_score_topk(int score_kind, int U, int n_local, int d, const float *users_tab,
                               const int32_t *user_ids, const float *items, const float *sig_u,
                               const float *sig_i, float c, const int32_t *mask_ptr, const int32_t *mask_idx,
                               int item_offset, int K, int n_splits, float *out_val, int32_t *out_idx,
                               void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(score_kind == MACR_SCORE_NORMAL || score_kind == MACR_SCORE_RUBI_BOTH, MACR_E_INVALID,
                 "score_topk: score_kind=%d", score_kind);
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_topk: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_topk: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "score_topk: K=%d outside [1,%d]", K, MACR_MAX_TOPK);
    MACR_REQUIRE(users_tab && items && out_val && out_idx, MACR_E_INVALID, "score_topk: null pointer");
    MACR_REQUIRE(score_kind == MACR_SCORE_NORMAL || (sig_u && sig_i), MACR_E_INVALID,
                 "score_topk: RUBI_BOTH needs sig_u and sig_i");
    MACR_REQUIRE((mask_ptr == nullptr) == (mask_idx == nullptr) || mask_ptr, MACR_E_INVALID, "score_topk: mask_idx without mask_ptr");
    if (n_splits <= 0) n_splits = macr_score_topk_splits(U, n_local, d);
    const int ublocks = (U + kUsersPerBlock - 1) / kUsersPerBlock;
    const size_t smem = score_topk_smem_bytes();
    hipStream_t st = as_stream(stream);
    uint32_t *shared_thr = nullptr;             // per-query admission bounds shared by the item splits
    if (workspace && n_splits > 1) {
        MACR_REQUIRE(workspace_bytes >= (size_t)U * 4, MACR_E_WORKSPACE, "score_topk: workspace %zu < %zu bytes",
                     workspace_bytes, (size_t)U * 4);
        shared_thr = static_cast<uint32_t *>(workspace);
        hipError_t me = hipMemsetAsync(shared_thr, 0, (size_t)U * 4, st);
        MACR_REQUIRE(me == hipSuccess, MACR_E_LAUNCH, "score_topk: memset: %s", hipGetErrorString(me));
    }
    MACR_DISPATCH_DK(d, score_kind, {
        auto kern = k_score_topk<D, KIND>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        MACR_REQUIRE(e == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
        kern<<<ublocks * n_splits, 512, smem, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c,
                                                    mask_ptr, mask_idx, item_offset, K, n_splits, out_val, out_idx,
                                                    shared_thr);
    });
    MACR_CHECK_LAUNCH("score_topk", st);
    return MACR_OK;
}

extern "C" int macr_score_matrix(int score_kind, int U, int n_local, int d, const float *users_tab,
                                 const int32_t *user_ids, const float *items, const float *sig_u,
                                 const float *sig_i, float c, float *out_scores, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(score_kind == MACR_SCORE_NORMAL || score_kind == MACR_SCORE_RUBI_BOTH, MACR_E_INVALID,
                 "score_matrix: score_kind=%d", score_kind);
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_matrix: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_matrix: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(users_tab && items && out_scores, MACR_E_INVALID, "score_matrix: null pointer");
    MACR_REQUIRE(score_kind == MACR_SCORE_NORMAL || (sig_u && sig_i), MACR_E_INVALID,
                 "score_matrix: RUBI_BOTH needs sig_u and sig_i");
    dim3 grid((n_local + 127) / 128, (U + 31) / 32);
    MACR_DISPATCH_DK(d, score_kind, (k_score_matrix<D, KIND><<<grid, 256, 0, st>>>(
                                        U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, out_scores)));
    MACR_CHECK_LAUNCH("score_matrix", st);
    return MACR_OK;
}

extern "C" int macr_topk_scores(const float *scores, int cols, int rows, int K, int32_t *out_idx,
                                float *out_val, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(scores && out_idx, MACR_E_INVALID, "topk_scores: null pointer");
    MACR_REQUIRE(cols > 0 && rows > 0, MACR_E_INVALID, "topk_scores: cols=%d rows=%d", cols, rows);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "topk_scores: K=%d outside [1,%d]", K, MACR_MAX_TOPK);
    k_topk_scores<<<(rows + 3) / 4, 256, 0, st>>>(scores, cols, rows, K, out_idx, out_val);
    MACR_CHECK_LAUNCH("topk_scores", st);
    return MACR_OK;
}

extern "C" int macr_topk_merge(int W, int U, int K, const float *vals, const int32_t *idxs,
                               const int32_t *fill_mask_ptr, const int32_t *fill_mask_idx, float *out_val,
                               int32_t *out_idx, int32_t *out_cnt, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(W >= 1 && U > 0, MACR_E_INVALID, "topk_merge: W=%d U=%d", W, U);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "topk_merge: K=%d outside [1,%d]", K, MACR_MAX_TOPK);
    MACR_REQUIRE(vals && idxs && out_val && out_idx, MACR_E_INVALID, "topk_merge: null pointer");
    k_topk_merge<<<(U + 3) / 4, 256, 0, st>>>(W, U, K, vals, idxs, fill_mask_ptr, fill_mask_idx,
                                                             out_val, out_idx, out_cnt);
    MACR_CHECK_LAUNCH("topk_merge", st);
    return MACR_OK;
}

extern "C" int macr_branch_sigmoid(const float *rows, const int32_t *idx, int n, int d, const float *w,
                                   float *out, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(rows && w && out, MACR_E_INVALID, "branch_sigmoid: null pointer");
    MACR_REQUIRE(n > 0, MACR_E_INVALID, "branch_sigmoid: n=%d", n);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "branch_sigmoid: d=%d not in {32,64,128,256}", d);
    MACR_DISPATCH_LPR(d, (k_branch_sigmoid<LPR><<<(n + (256 / LPR) - 1) / (256 / LPR), 256, 0, st>>>(
                             rows, idx, n, w, out)));
    MACR_CHECK_LAUNCH("branch_sigmoid", st);
    return MACR_OK;
}

extern "C" int macr_metrics_foldout(int U, int K, const int32_t *rankings, const int32_t *gt_ptr,
                                    const int32_t *gt_idx, float *results, int hr_in_ap_slot, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && K >= 1, MACR_E_INVALID, "metrics_foldout: U=%d K=%d", U, K);
    MACR_REQUIRE(rankings && gt_ptr && gt_idx && results, MACR_E_INVALID, "metrics_foldout: null pointer");
    k_metrics_foldout<<<(U + 127) / 128, 128, 0, st>>>(U, K, rankings, gt_ptr, gt_idx, results, hr_in_ap_slot);
    MACR_CHECK_LAUNCH("metrics_foldout", st);
    return MACR_OK;
}

extern "C" int macr_metrics_mf(int U, int Kmax, const int32_t *rankings, const int32_t *cnt,
                               const int32_t *gt_ptr, const int32_t *gt_idx, const int32_t *Ks, int nK,
                               double *out, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && Kmax >= 1, MACR_E_INVALID, "metrics_mf: U=%d Kmax=%d", U, Kmax);
    MACR_REQUIRE(rankings && gt_ptr && gt_idx && Ks && out, MACR_E_INVALID, "metrics_mf: null pointer");
    MACR_REQUIRE(nK >= 1 && nK <= 8, MACR_E_UNSUPPORTED, "metrics_mf: nK=%d outside [1,8]", nK);
    KsArg ka;
    ka.n = nK;
    for (int q = 0; q < nK; ++q) {
        MACR_REQUIRE(Ks[q] >= 1 && Ks[q] <= Kmax, MACR_E_INVALID, "metrics_mf: Ks[%d]=%d outside [1,%d]", q, Ks[q], Kmax);
        ka.k[q] = Ks[q];
    }
    k_metrics_mf<<<(U + 127) / 128, 128, 0, st>>>(U, Kmax, rankings, cnt, gt_ptr, gt_idx, ka, out);
    MACR_CHECK_LAUNCH("metrics_mf", st);
    return MACR_OK;
}

extern "C" int macr_colmean(const void *in, int in_is_f32, int rows, int cols, double *out, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(in && out && rows > 0 && cols > 0, MACR_E_INVALID, "colmean: bad arguments");
    if (in_is_f32)
        k_colmean<float><<<cols, 256, 0, st>>>(static_cast<const float *>(in), rows, cols, out);
    else
        k_colmean<double><<<cols, 256, 0, st>>>(static_cast<const double *>(in), rows, cols, out);
    MACR_CHECK_LAUNCH("colmean", st);
    return MACR_OK;
}
