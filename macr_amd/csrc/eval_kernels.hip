// Full-catalogue evaluator kernels for gfx950 (MI355X).
//
//   k_score_stream U.I^T on the fp32 MFMA (v_mfma_f32_32x32x2_f32) with the
//   k_tau          (y - c)*sig_i*sig_u epilogue and train-item masking, as a sampling pass
//   k_select       (per-user threshold) and a listing pass (candidates above it), plus the
//                  exact top-K selection -- the (U,N) score matrix never exists.
//   k_score_topk   the same ranking with a running per-user top-K in LDS: the first design,
//                  kept as the fallback that is exact for any score order.
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

#include <algorithm>

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
constexpr int kSelRegs = 16;                       // k_select ranks up to 64*kSelRegs = 1024 candidates per user (more: repair round)

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
// keeps the best K by a radix select on lane masks (compact_buffer).
// Since ABI 5 this kernel is the FALLBACK of macr_score_topk: launched after the streaming
// passes, it returns at once unless *run_flag says a candidate list overflowed.
// ----------------------------------------------------------------------------
template <int D, int KIND>
__global__ __launch_bounds__(512, 2) void k_score_topk(
    int U, int n_local, const float *__restrict__ users_tab, const int32_t *__restrict__ user_ids,
    const float *__restrict__ items, const float *__restrict__ sig_u, const float *__restrict__ sig_i, float c_val,
    const float *__restrict__ c_dev,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_idx, int item_offset, int K,
    int n_splits, float *__restrict__ out_val, int32_t *__restrict__ out_idx, uint32_t *shared_thr,
    const int32_t *run_flag, int32_t *stats, const int32_t *__restrict__ only_blocks = nullptr) {
    // run_flag[0]: a candidate list overflowed twice (this kernel must run); run_flag[1]: user blocks the repair round re-listed
    if (stats && run_flag && blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = run_flag[1]; stats[1] = run_flag[0]; }
    if (run_flag && *run_flag == 0) return;             // fallback launch: only when a candidate list overflowed
    // ... and then only for the blocks of 256 queries the repair round listed again (k_repair_plan: blk_flag != 0): a second
    // overflow can only be among them, every other block's first-round ranking stands.  (One degenerate query -- a user
    // whose branch factor has collapsed every score to a tie -- used to send a whole 100 000-query evaluation through
    // this kernel: 445 ms instead of 87 at the configs[4] shape.)
    if (run_flag && only_blocks && only_blocks[blockIdx.x / n_splits] == 0) return;
    const float c = c_dev ? *c_dev : c_val;             // device-resident c: one captured graph serves a whole c sweep
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
    const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;

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
        return (score_uses_sig_i(KIND) && tid < kTileItems && it < it_hi) ? sig_i[it] : 0.f;
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
            if (score_uses_sig_i(KIND)) v = score_epilogue<KIND>(v, c, s_sig[sbuf * kTileItems + il], su);
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
// k_score_stream + k_tau + k_select: the full-catalogue ranking as a FIXED-threshold stream.
//
// On gfx950 the fp32-input MFMA and the fp32 VALU do not overlap (tools/mfma_valu_bench.hip: every
// VALU instruction beside v_mfma_f32_32x32x2_f32 costs ~3 cycles on top of the MFMA's 64), so the
// cost of a ranking kernel is MFMA time + everything else, and a running top-K per user (buffers,
// compaction, thresholds that move) was 2/3 of k_score_topk.  Here the threshold of a user is
// CONSTANT during a launch and candidates are only appended:
//   pass 0 (MODE_MAX)   every 8th item tile.  Nothing is listed: each lane keeps the running maximum of each of
//                       its 16 accumulator slots (one v_max per score).  A (split, lane half, slot) class holds
//                       distinct items, so the K-th largest of a user's S*32 class maxima is the score of at
//                       least K distinct items: a valid lower bound tau of the user's K-th best score, at about
//                       rank K in the 1/8 sample, i.e. rank ~8K overall -- whatever the catalogue size.
//   k_tau               one wave per user: tau = K-th largest class maximum.
//   pass 1 (MODE_LIST)  all tiles: an item is listed iff score >= tau (~8K per user); per element that is the
//                       score epilogue and one compare, per listed item an LDS counter bump and an 8-byte store.
//   k_select            one wave per user: exact top K of the list by (score desc, id asc), sorted.
// No key buffers in LDS: 18 KB per block instead of 152 KB, so two blocks (4 waves per SIMD) share a CU
// and hide each other's barriers and LDS waits.  Pass 0 re-multiplies 1/8 of the tiles; that is cheaper
// than any way of listing the sample.
// Exactness: the list of a user always contains her true top K (tau is a lower bound of the K-th best
// score, ties are listed); scores are the same MFMA accumulation and epilogue as k_score_topk / the
// oracle.  A list that overflows (adversarial score order) raises *overflow and the launch sequence falls
// back to k_score_topk, which is exact for any input.
//
// Masked (train) items and the tail of the last tile are poisoned at accumulator INIT (NaN + x = NaN,
// NaN >= tau is false, max(m, NaN) = m): nothing per element.  Which items of a tile a user masks is one
// 32-bit word of a (tile, user) bitmap that k_mask_bits scatters from the CSR lists once per call; the
// word is prefetched with the tile (a CSR cursor costs a dependent load whenever it advances).
// A operand in LDS as [item][h][t] (k = 2t+h), row stride D+4: a lane's whole k-row is contiguous
// (ds_read_b128) and 16 lanes x 16 B cover all 64 banks once.
// ----------------------------------------------------------------------------
// ---- repair round: where the re-listed query blocks keep their lists ---------------------------------------------------
// A query block spreads over as many blocks of the grid as it has result slots, and the workspace has slots_full of them
// per user: listed again alone, a query block would keep ~1/slots_full of the grid busy for a whole pass.  The lists of
// everybody else are dead by then (the first selection has read them), so the n_ub re-listed query blocks take the whole
// buffer: compact user index (position in ub_map) * 256 + slot, stride n_ub * 256, and slots_full * U / (n_ub * 256)
// (<= 64) slots each -- one re-listed query block of the Gowalla shape runs on 63 blocks of the grid instead of 8.
// k_repair_plan stores position + 1 in blk_flag and clears the counts of that layout.
struct RepairLayout { bool compact; int slots, stride; };
__device__ __forceinline__ RepairLayout repair_layout(int slots_full, int U, int n_ub) {
    RepairLayout r;
    long long sl = (long long)slots_full * U / ((long long)n_ub * kUsersPerBlock);
    if (sl > 64) sl = 64;
    r.compact = sl > slots_full;
    r.slots = r.compact ? (int)sl : slots_full;
    r.stride = r.compact ? n_ub * kUsersPerBlock : U;
    return r;
}
// blocks of the grid a repair round uses for n_ub query blocks of T visits each
__device__ __forceinline__ long long repair_grid(const RepairLayout &r, long long grid, int ublocks, int n_ub, int T) {
    long long G;
    if (r.compact) {
        long long chunk = (T + r.slots - 2) / (r.slots - 1);          // a query block then overlaps <= slots ranges
        if (chunk < 8) chunk = 8;
        G = (long long)n_ub * T / chunk;
        if (G > grid) G = grid;
    } else {
        G = (long long)n_ub * grid / ublocks;                        // ranges as long as in the full launch
    }
    return G < 1 ? 1 : G;
}

constexpr int kModeMax = 0, kModeList = 1;
// Pass 0 is a SAMPLE of the catalogue: one "virtual tile" per window of 2^s consecutive tiles, made of every 2^s-th
// item of the window (items w*32*2^s + 2^s*j + phase, j = 0..31, phase = w % 2^s) -- a 1/2^s sample spread evenly
// over the item ids; s = 3.  It used to be every 8th TILE (32 consecutive items): cheaper to stage, but item ids
// follow popularity, a trained model's best items sit in a few adjacent tiles, and a tile-granular sample either
// holds most of a user's top items or none of them (measured after ~800 training steps with a 1/16 tile sample: tau
// so low that the candidate lists overflowed, profiles/r02a_kernel_stats.csv).  The strided sample has the variance
// of a uniform one: the rank of the K-th best sampled item is K*2^s +- sqrt(K)*2^s = 160 +- 36 for K = 20.
// s = 4 was measured with the strided sample too (Gowalla shape): the sampling pass drops from 103 to 67 us, but the
// lists double (rank 320 +- 70), k_select needs 8 keys per lane instead of 4 (22 -> 42 us; ML-10M shape, 70 k users:
// 23 -> 56 us) and on a model whose users all rank the catalogue alike the candidates of a user land in ONE tile
// range and a few users run over the 512-entry list of that range: 1-3 query blocks re-listed per evaluation.
// Net gain 16 us at best: every 8th item stays.  The train-item mask of a virtual tile is one word of a second
// bitmap that k_mask_bits builds beside the per-tile one.  -DMACR_SAMPLE_LOG2=n rebuilds with another rate.
static inline int sample_log2(int n_local) {
#ifdef MACR_SAMPLE_LOG2
    return MACR_SAMPLE_LOG2;
#else
    (void)n_local;
    return 3;
#endif
}
static inline int n_tiles(int n_local) { return (n_local + kTileItems - 1) / kTileItems; }
static inline int n_windows(int n_local) { const int sl = sample_log2(n_local); return (n_tiles(n_local) + (1 << sl) - 1) >> sl; }

template <int D>
struct StreamCfg {
    static constexpr int RS = D + 4;                 // LDS row stride (floats)
    static constexpr int NT = D / 2;                 // MFMA steps per tile
    static constexpr int LD4 = (kTileItems * D / 4 + 511) / 512;   // float4 per thread per tile
    static constexpr size_t smem = (size_t)2 * kTileItems * RS * 4 + 2 * kTileItems * 4 + kUsersPerBlock * 4;
    static constexpr size_t smem_sweep = smem + (size_t)(4 - 1) * kUsersPerBlock * 4;      // one counter array per c (kMaxSweep)
};

// Initialises the head of the ranking workspace in ONE launch: [0, n_zero) words <- 0 (counts, flags), the next n_tau
// words <- tau_bits when set_tau (the -inf thresholds of the list-everything path), the next n_max words <- all-ones
// (NaN: "this class saw nothing").  A kernel, not hipMemsetAsync: memset nodes of a captured HIP graph lose their
// effect from the second replay on (ROCm 7.2, tools/graph_memset_check.py), and the evaluator replays this sequence.
__global__ __launch_bounds__(256) void k_topk_ws_init(uint32_t *__restrict__ base, size_t n_zero, size_t n_tau,
                                                      uint32_t tau_bits, int set_tau, size_t n_max) {
    const size_t total = n_zero + n_tau + n_max;
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < total; w += (size_t)gridDim.x * blockDim.x) {
        if (w < n_zero) base[w] = 0u;
        else if (w < n_zero + n_tau) { if (set_tau) base[w] = tau_bits; }
        else base[w] = 0xffffffffu;
    }
}

// mask_bits[tile][user] |= 1 << (item % 32) for every masked item of the shard, and the same for the sampling pass's
// virtual tiles (sample_log2); one wave per user.
__global__ __launch_bounds__(256) void k_mask_bits(int U, int n_local, const int32_t *__restrict__ mask_ptr,
                                                   const int32_t *__restrict__ mask_idx, int item_offset,
                                                   uint32_t *__restrict__ mask_bits, int tiles, int sl) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= U) return;
    const int e0 = mask_ptr[q], e1 = mask_ptr[q + 1];
    uint32_t *sample_bits = mask_bits + (size_t)tiles * U;       // [window][user]: bit j = item j of the window's virtual tile
    for (int e = e0 + lane; e < e1; e += 64) {
        const int it = mask_idx[e] - item_offset;
        if (it >= 0 && it < n_local) {
            atomicOr(&mask_bits[(size_t)(it >> 5) * U + q], 1u << (it & 31));
            const int w = it >> (5 + sl), rem = it - (w << (5 + sl));
            if ((rem & ((1 << sl) - 1)) == (w & ((1 << sl) - 1))) atomicOr(&sample_bits[(size_t)w * U + q], 1u << (rem >> sl));
        }
    }
}

// c sweep (macr_score_topk_sweep): ONE listing pass serves up to kMaxSweep values of c -- c only enters the epilogue, so
// the MFMA work, the staging of the item tiles and the barriers are shared, and per (score, c) there is one epilogue,
// one compare and (for the ~8K listed items per user and c) one append into that c's own lists.
constexpr int kMaxSweep = 4;
struct SweepArgs {
    int n_c;                                  // values of c in this launch (<= kMaxSweep)
    const float *tau[kMaxSweep];              // [U] per c
    uint64_t *lists[kMaxSweep];               // [slots][U][cap] per c
    int32_t *counts[kMaxSweep];               // [slots][U] per c
    int32_t *overflow[kMaxSweep];             // one flag per c
};

template <int D, int KIND, int MODE, int NC = 1, bool REPAIR = false>
__global__ __launch_bounds__(512, D <= 64 ? 4 : 2) void k_score_stream(
    int U, int n_local, const float *__restrict__ users_tab, const int32_t *__restrict__ user_ids,
    const float *__restrict__ items, const float *__restrict__ sig_u, const float *__restrict__ sig_i, float c_val,
    const float *__restrict__ c_dev,
    const uint32_t *__restrict__ mask_bits, int item_offset,
    int ublocks, const float *__restrict__ tau, float *__restrict__ maxima, uint64_t *__restrict__ lists,
    int32_t *__restrict__ counts, int cap, int32_t *overflow, int ovf_per_user, int sample_log2,
    const int32_t *__restrict__ ub_map, const int32_t *__restrict__ n_ub_dev, int32_t *blk_flag, SweepArgs sw,
    int slots_full) {
    using C = StreamCfg<D>;
    // Seeded first round (blk_flag != NULL): "these seeds are stale" is decided early.  2, 8 and 32 tiles into a user
    // block's range every wave compares what its 32 users have listed with what a usable threshold lists (a projected
    // kSelRegs*64 candidates per user over the shard is the most k_select can rank); one wave over that and the block
    // stops listing and marks the user block in blk_flag, its sibling blocks (same user block, other tile ranges) see
    // the mark within 8 tiles and stop too, and the repair round (sampling pass, k_tau, listing, selection) takes the
    // user block over: stale seeds waste ~6 % of a listing pass instead of all of it.
    constexpr bool kEarlyStop = MODE == kModeList && NC == 1 && !REPAIR;
    constexpr int kCheckTiles = 8;
    static_assert(NC == 1 || MODE == kModeList, "the sweep shares the listing pass only");
    const float c = c_dev ? *c_dev : c_val;
    constexpr int RS = C::RS, NT = C::NT;
    const int kStep = MODE == kModeMax ? (1 << sample_log2) : 1;
    extern __shared__ __align__(16) unsigned char smem[];
    float *s_a = reinterpret_cast<float *>(smem);                       // [2][32][RS]
    float *s_sig = s_a + 2 * kTileItems * RS;                           // [2][32]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_sig + 2 * kTileItems);   // [NC][256]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int uslot = wid * 32 + col;
    const int tiles_total = (n_local + kTileItems - 1) / kTileItems;

    // Work = (user block, visited tile) pairs in user-block-major order, cut into gridDim.x equal ranges: every
    // resident block slot gets the same number of tiles whatever ublocks is.  A block's range may end one user
    // block and begin the next ("segments"); the k-th block overlapping a user block writes that block's
    // result slot k (lists / counts / maxima are [slot][user]).
    const int T = (tiles_total + kStep - 1) / kStep;                    // visited tiles per user block
    // Repair round (ub_map != NULL): only the *n_ub_dev user blocks of ub_map are listed again, by as many blocks of the
    // grid as keep a block's tile range at least as long as in the full launch (so a user block never spans more
    // result slots than the workspace has); the others leave at once.
    int n_ub = ublocks;
    long long G = gridDim.x;
    const long long b = blockIdx.x;
    RepairLayout rl = {false, 0, U};
    if (REPAIR) {                                  // (an instantiation of its own: the profiler tells the rounds apart)
        n_ub = *n_ub_dev;
        if (n_ub == 0) return;
        if (MODE == kModeList) rl = repair_layout(slots_full, U, n_ub);       // (the sampling pass keeps the layout of its maxima)
        G = repair_grid(rl, G, ublocks, n_ub, T);
        if (b >= G) return;
    }
    // A block's range [i0, i1) of VISITED tiles is not a contiguous stretch of the catalogue: visit i is tile (i * S) mod T,
    // S coprime to T near 0.618 T.  Item ids follow popularity in recommender data, a trained model's best items sit in a
    // few adjacent tiles, and with contiguous ranges ONE block's 32 threshold classes held nearly all of a user's top
    // items of the sampling pass: the K-th largest class maximum then sits far below the K-th best sampled score, the
    // lists grow several-fold and the listing pass with them (measured on the bench's model after 54 000 training steps:
    // 698 -> 876 us).  Scattered, every block's classes see the same mix of popular and unpopular tiles.
    int S = (int)(0.6180339f * (float)T);
    S = S < 1 ? 1 : S;
    for (;; ++S) {                                                      // (wave-uniform, a handful of iterations)
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T) * kStep; };
    // visit(i + 1) from visit(i): no 64-bit modulo per visit (kStep is a power of two, a visit a multiple of it)
    auto visit_after = [&](int tile) { const int n = tile / kStep + S; return (n >= T ? n - T : n) * kStep; };
    const long long W = (long long)n_ub * T;
    const long long w_end = W * (b + 1) / G;
    for (long long w = W * b / G; w < w_end;) {
    const int ubv = (int)(w / T), i0 = (int)(w - (long long)ubv * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ubv * T * G / W;                       // the block holding this user block's first tile
    while (W * (first + 1) / G <= (long long)ubv * T) ++first;
    while (W * first / G > (long long)ubv * T) --first;
    const int split = (int)(b - first);
    const int ub = REPAIR ? ub_map[ubv] : ubv;
    const int q = ub * kUsersPerBlock + uslot;
    const bool q_ok = q < U;

    if (MODE == kModeList && tid < kUsersPerBlock) {
#pragma unroll
        for (int g = 0; g < NC; ++g) s_cnt[g * kUsersPerBlock + tid] = 0u;
    }
    float bfrag[NT];
    {
        const float *urow = users_tab + (size_t)(q_ok ? (user_ids ? user_ids[q] : q) : 0) * D;
#pragma unroll
        for (int t4 = 0; t4 < D / 4; ++t4) {           // k = 4*t4 .. 4*t4+3  ->  steps 2*t4 (k=+0,+1) and 2*t4+1 (k=+2,+3)
            const float4 v = q_ok ? ld4(urow + 4 * t4) : make_float4(0.f, 0.f, 0.f, 0.f);
            bfrag[2 * t4] = h ? v.y : v.x;
            bfrag[2 * t4 + 1] = h ? v.w : v.z;
        }
    }
    const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;
    // listing test: score >= tau_s (NaN = never: padding users, poisoned scores)
    float tau_s = (MODE == kModeList && NC == 1 && q_ok) ? tau[q] : __builtin_nanf("");       // +inf once her list is full
    // (repair round, compact layout: the lists are indexed by the query block's position among the re-listed ones)
    const int ql = (REPAIR && rl.compact) ? ubv * kUsersPerBlock + uslot : (q_ok ? q : 0);
    uint64_t *my_list = lists + ((size_t)split * rl.stride + ql) * cap;
    float tau_g[NC], c_g[NC];                 // sweep: per-c threshold and constant (NC > 1)
    if (NC > 1) {
#pragma unroll
        for (int g = 0; g < NC; ++g) {
            tau_g[g] = (g < sw.n_c && q_ok) ? sw.tau[g][q] : __builtin_nanf("");
            c_g[g] = g < sw.n_c ? c_dev[g] : 0.f;
        }
    }

    int vi = i0, t = visit(i0);
    // staging: thread -> (item row, float4 column); k=4c..4c+3 lands as (h=0: t=2c,2c+1 <- x,z) (h=1: <- y,w)
    float4 stg[C::LD4];
    float sg = 0.f;
    uint32_t tm_next = 0u;
    // Rows past the end of the shard are clamped, not zeroed: their scores are poisoned through tmask anyway, and
    // an unconditional load lets the compiler keep the prefetch in flight across the whole MFMA phase.
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int k = 0; k < C::LD4; ++k) {
            const int e = tid + 512 * k, row = (e / (D / 4)) & (kTileItems - 1), c4 = e % (D / 4);
            // sampling pass: `tile` is the first tile of a window, item j of the virtual tile = every kStep-th of the window
            const int it = MODE == kModeMax ? min(tile * kTileItems + kStep * row + ((tile >> sample_log2) & (kStep - 1)), n_local - 1)
                                            : min(tile * kTileItems + row, n_local - 1);
            stg[k] = ld4(items + (size_t)it * D + 4 * c4);
        }
        if (score_uses_sig_i(KIND)) {
            const int row = tid & (kTileItems - 1);
            sg = sig_i[MODE == kModeMax ? min(tile * kTileItems + kStep * row + ((tile >> sample_log2) & (kStep - 1)), n_local - 1)
                                        : min(tile * kTileItems + row, n_local - 1)];
        }
        // masked items of (tile, user); sampling pass: of (window, user), the second bitmap behind the per-tile one
        if (MODE == kModeMax) tm_next = (mask_bits && q_ok) ? mask_bits[((size_t)tiles_total + (tile >> sample_log2)) * U + q] : 0u;
        else tm_next = (mask_bits && q_ok) ? mask_bits[(size_t)tile * U + q] : 0u;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int k = 0; k < C::LD4; ++k) {
            // opaque to the optimiser: otherwise the (x,z)/(y,w) register shuffle -- and with it the wait for the
            // prefetch -- is hoisted to right after the load, in front of the MFMA phase
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            const int e = tid + 512 * k, row = e / (D / 4), c4 = e % (D / 4);
            if (row < kTileItems) {
                float *p = s_a + ((size_t)buf * kTileItems + row) * RS + 2 * c4;
                *reinterpret_cast<float2 *>(p) = make_float2(stg[k].x, stg[k].z);
                *reinterpret_cast<float2 *>(p + NT) = make_float2(stg[k].y, stg[k].w);
            }
        }
        if (score_uses_sig_i(KIND)) {
            asm volatile("" : "+v"(sg));
            if (tid < kTileItems) s_sig[buf * kTileItems + tid] = sg;
        }
    };

    float cmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cmax[r] = -INFINITY;
    int buf = 0;
    if (vi < i1) { load_tile(t); store_tile(0); }
    uint32_t tm_cur = tm_next;
    __syncthreads();
    const float kNone = __builtin_nanf("");
#ifdef MACR_ABL_S_NOLOOP
    vi = i1;
#endif
    while (vi < i1) {
        const bool has_next = vi + 1 < i1;
        const int tn = has_next ? visit_after(t) : t;
        const uint32_t tm_this = tm_cur;
        (void)tm_this;
#ifndef MACR_ABL_S_NOSTAGE
        if (has_next) load_tile(tn);
#endif

        const int gid0 = t * kTileItems + item_offset;
        uint32_t tmask = tm_cur;
#ifdef MACR_ABL_S_NOMASK
        tmask = 0;
#endif
        // < 32 only in the last tile (window) of the shard
        const int valid = MODE == kModeMax ? (n_local - t * kTileItems - ((t >> sample_log2) & (kStep - 1)) + kStep - 1) >> sample_log2
                                           : n_local - t * kTileItems;
        if (valid < kTileItems) tmask |= valid > 0 ? ~0u << valid : ~0u;

        const float *ua = s_a + ((size_t)buf * kTileItems + col) * RS + h * NT;
        f32x16 acc;
        float a4[4];
        *reinterpret_cast<float4 *>(a4) = *reinterpret_cast<const float4 *>(ua);
        if (__any(tmask != 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = ((tmask >> ((r & 3) + 8 * (r >> 2) + 4 * h)) & 1u) ? kNone : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], bfrag[0], acc, 0, 0, 0);
        } else {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], bfrag[0], zero, 0, 0, 0);
        }
#ifndef MACR_ABL_S_NOMFMA
#pragma unroll
        for (int tt = 1; tt < NT; ++tt) {
            if ((tt & 3) == 0) *reinterpret_cast<float4 *>(a4) = *reinterpret_cast<const float4 *>(ua + tt);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tt & 3], bfrag[tt], acc, 0, 0, 0);
        }
#endif

        // epilogue: 16 scores per lane
        float sgi[16];
        if (score_uses_sig_i(KIND)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(sgi + 4 * g) = *reinterpret_cast<const float4 *>(s_sig + buf * kTileItems + 8 * g + 4 * h);
        }
#ifdef MACR_ABL_S_NOEPI
        if (acc[0] == 12345.f) *overflow = 1;
#else
        float v[16];
        uint64_t hit = 0ull;                                // OR of the compare masks: scalar unit, not VALU
        if (NC > 1) {
            // one epilogue + compare per (score, c); appends go to the lists of that c
#pragma unroll
            for (int g = 0; g < NC; ++g) {
                if (g >= sw.n_c) break;                     // wave-uniform
                uint64_t hg = 0ull;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = score_epilogue<KIND>(acc[r], c_g[g], sgi[r], su);
                    hg |= __ballot(v[r] >= tau_g[g]);
                }
                if (hg) {
                    uint64_t *lst = sw.lists[g] + ((size_t)split * U + (q_ok ? q : 0)) * cap;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (v[r] >= tau_g[g]) {
                            const uint32_t pos = atomicAdd(&s_cnt[g * kUsersPerBlock + uslot], 1u);
                            if (pos < (uint32_t)cap) lst[pos] = make_key(v[r], gid0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                            else *sw.overflow[g] = 1;
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = acc[r];
            if (score_uses_sig_i(KIND)) v[r] = score_epilogue<KIND>(v[r], c, sgi[r], su);
            if (MODE == kModeMax) cmax[r] = fmaxf(cmax[r], v[r]);
            else hit |= __ballot(v[r] >= tau_s);
        }
        }
#ifdef MACR_ABL_S_NOAPPEND
        if (v[3] == 12345.f) *overflow = 1;
        hit = 0ull;
#endif
        if (MODE == kModeList && NC == 1 && hit) {          // one wave-uniform branch per tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (v[r] >= tau_s) {
                    const uint32_t pos = atomicAdd(&s_cnt[uslot], 1u);
                    if (pos < (uint32_t)cap) my_list[pos] = make_key(v[r], gid0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                    else { overflow[ovf_per_user ? q : 0] = 1; tau_s = INFINITY; }     // full: stop listing for her
                }
            }
        }
#endif
        bool stop = false;                                   // block-uniform
        if (kEarlyStop && blk_flag) {
            const int done = vi - i0 + 1;                    // tiles of this range processed
#ifdef MACR_ABL_STOPNOW
            const bool check = done == MACR_ABL_STOPNOW;
#else
            const bool check = done == 2 || done == kCheckTiles || done == 4 * kCheckTiles;
#endif
            if (check || (done & 7) == 0) {
                bool mine = false;
                if (check) {
                    uint32_t a = h == 0 ? s_cnt[uslot] : 0u; // appended so far by this wave's 32 users
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, kWave);
                    const float usable = 32.f * (float)(done * kTileItems) * (float)(kSelRegs * 64) / (float)n_local;
                    mine = (float)a > usable + 64.f;
#if defined(MACR_ABL_STOPNOW) && !defined(MACR_ABL_STOPCRIT)
                    mine = true;
#endif
                } else if (tid == 0) {                       // a sibling block (same users, other tiles) gave up
                    mine = __hip_atomic_load(blk_flag + ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                }
                stop = __syncthreads_or(mine) != 0;          // (a barrier of its own: every wave gets the same answer)
                // agent-scope store: the sibling blocks run on other XCDs, whose L2 a plain store would not reach before the kernel ends
                if (stop && tid == 0) __hip_atomic_store(blk_flag + ub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifndef MACR_ABL_S_NOSTAGE
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
#endif
        tm_cur = tm_next;
        buf ^= 1;
        t = tn; ++vi;
        if (kEarlyStop && stop) break;
    }
    if (MODE == kModeMax) {
        if (q_ok) {
            float *o = maxima + ((size_t)split * U + q) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g)       // slots (r&3)+8(r>>2)+4h: four runs of four consecutive floats
                *reinterpret_cast<float4 *>(o + 8 * g + 4 * h) = make_float4(cmax[4 * g], cmax[4 * g + 1], cmax[4 * g + 2], cmax[4 * g + 3]);
        }
    } else if (tid < kUsersPerBlock) {
        const int qq = ub * kUsersPerBlock + tid;
        if (NC == 1) {
            const int ql2 = (REPAIR && rl.compact) ? ubv * kUsersPerBlock + tid : qq;
            if (qq < U) counts[(size_t)split * rl.stride + ql2] = (int32_t)min(s_cnt[tid], (uint32_t)cap);
        } else {
#pragma unroll
            for (int g = 0; g < NC; ++g)
                if (g < sw.n_c && qq < U)
                    sw.counts[g][(size_t)split * U + qq] = (int32_t)min(s_cnt[g * kUsersPerBlock + tid], (uint32_t)cap);
        }
    }
    }   // segments
}

// ----------------------------------------------------------------------------
// k_tau_seed: thresholds from SEED items instead of a sampling pass.  Any K distinct unmasked items of a user give a
// valid lower bound of her K-th best score: the smallest of their K scores.  An evaluator ranks the same users against
// slowly moving tables epoch after epoch, so the best items of the PREVIOUS evaluation are nearly the best now.  A
// user has kSeedWidth = 32 seeds (the top 32 candidates k_select saw last time, more than K so that some of them may
// have dropped out of the top since): tau = the K-th largest of their exact current scores -- 32 dot products per
// user, one lane each, instead of the sampling pass and k_tau (112 + 27 us on the Gowalla shape), with a threshold
// at rank ~K instead of ~8K (shorter candidate lists, cheaper selection).
// Exactness: the dot product is the k-ascending fmaf chain of the MFMA kernels and the epilogue the same function, so
// a seed item's score here is bit-identical to its score in the listing pass and passes `score >= tau`.
// Seeds are checked, not trusted: an id that is -1 / outside this shard / masked / a repetition of an earlier lane's
// id / scores NaN does not count; a user with fewer than K good seeds gets tau = -inf (everything unmasked is listed:
// fewer than K items for users who never had K candidates, the repair round for anyone else).
// One 32-lane half per user, lane l = seed l.
// ----------------------------------------------------------------------------

// ============================================================================
// bf16 FILTER for the listing pass (opt-in: MACR_EVAL_FILTER=bf16 / macr_set_eval_filter).
// The listing pass only has to find, per user, a superset of her best K items; the ranking is decided on exact fp32
// scores.  So the (U, N) product may run on the bf16 matrix cores -- 16x the fp32 MFMA rate -- as long as nothing that
// belongs to the exact top K is lost:
//   * every operand as TWO bf16 numbers, x ~ hi + lo (hi = bf16(x), lo = bf16(x - hi), round to nearest even:
//     |x - hi - lo| <= 2^-16 |x|; k_bf16_prep), the product as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (products
//     exact, fp32 accumulation): 3 MFMAs of 32 cycles per 16 k instead of 8 of 64 -- 5.3x fewer matrix-core cycles --
//     and |s_bf16 - s_fp32| <= filter_rel(d) * sum_k |u_k q_k| <= filter_rel(d) * |u| * max_i |q_i|  (Cauchy-Schwarz; filter_rel below).
//     (One bf16 per operand -- 16x -- was built first: its bound, 2^-7 |u| max|q|, is wider than the gap between a user's
//     K-th and 64th best score whenever a few popular items have long rows, and most users failed the check below.)
//     every epilogue (common.hpp score_epilogue) is 1-Lipschitz in s (sigmoids <= 1) and adds <= 3 roundings of values
//     bounded by |s| + |c|.  margin_u = filter_margin(|u|, max|q|, c) bounds |score_bf16 - score_fp32| for every item;
//   * the pass lists every item with score_bf16 >= tau_u - margin_u (tau_u a valid fp32 threshold: >= K unmasked items
//     score >= tau_u), i.e. every item whose fp32 score is >= tau_u;
//   * k_select_b takes the 64 best listed items by bf16 score, checks that the 64th lies more than 2 margin_u below the
//     K-th -- then no item outside the 64 can belong to the exact top K -- computes the fp32 score of those within
//     2 margin_u of the K-th with the k-ascending fmaf chain and the exact epilogue of every other kernel here, and ranks them.
//     A user for whom the check fails (dozens of scores inside a margin) is flagged like a user whose list overflowed:
//     the repair round lists her user block again with the fp32 kernels.
// The result is the fp32 ranking, bit for bit; tests compare both filters with the oracle.
// ============================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
// Relative term of the margin, per embedding width d (a bound of |s_bf16 - s_fp32| / sum_k |u_k q_k|):
//   3.2 * 2^-16   the three dropped terms of the split product, lo*lo + r_u*q + u*r_q  (|lo| <= 2^-8 |x|, |r| <= 2^-16 |x|;
//                 3 * 2^-16 plus their second-order terms)
//   8 d * 2^-24   the accumulation: 3d exact bf16 products added into an fp32 accumulator over 3d/16 MFMAs -- <= 2^-23 per
//                 added term relative to the running sum of magnitudes whether the matrix core rounds to nearest or
//                 truncates the aligned addends (6 d 2^-24), the 17-term alignment of one instruction (< d 2^-24) -- and the
//                 fp32 fmaf chain of the exact score itself (d 2^-24).
// d = 32: 6.4e-5, 64: 7.9e-5, 128: 1.10e-4, 256: 1.71e-4.  tests/test_gpu_ops.py measures the element-wise error of the
// very MFMA sequence of the listing pass on adversarial operands (macr_test_bf16_products) against this margin.
__host__ __device__ constexpr float filter_rel(int d) { return 3.2f / 65536.f + 8.f * (float)d / 16777216.f; }
constexpr float kFilterAbs = 1.0e-6f;        // > 2^-20: roundings of the epilogue / of the test on the raw product, relative to |u| max|q| + |c|

__device__ __forceinline__ float filter_margin(int d, float unorm, float qmax, float c) {
    const float b = unorm * qmax * 1.0001f;
    return filter_rel(d) * b + kFilterAbs * (b + fabsf(c)) + 1e-37f;
}

// ----------------------------------------------------------------------------
// fp16 FILTER (MACR_EVAL_FILTER_F16, round 6): ONE fp16 number per operand, one MFMA per 16 k instead of three.
// fp16 keeps 11 significant bits where bf16 keeps 8, so x ~ fp16(x) alone (round to nearest even) is 8x closer than one
// bf16 and the margin -- 2^-10 |u| max|q| -- stays several times below the gap between a query's K-th and 64th best score
// (which the one-bf16 filter's 2^-7 did not: see above); what the wider margin costs is a few more exact re-scores in
// k_select_b.  Everything else is the bf16 filter's: thresholds minus margin, 64 best by filter score, fp32 re-scoring,
// per-query fall-back -- the fp32 ranking bit for bit.  The bound, element-wise:
//   * |x - fp16(x)| <= 2^-11 |x| for |x| >= 2^-14, <= 2^-25 below (fp16 subnormals; v_mfma_f32_32x32x16_f16 does NOT flush
//     them: tools/f16_mfma_check.hip, and the element-wise test runs operands over 24 binades), so
//     sum_k |u_k q_k - u^_k q^_k| <= (2^-10 + 2^-22) sum_k |u_k q_k| + 2^-25 (1 + 2^-11)(|u|_1 + |q|_1) + d 2^-50
//                                 <= 1.0005 * 2^-10 |u| max|q| + 2^-25 * 1.001 sqrt(d) (|u| + max|q|);
//   * products of two fp16 numbers are exact in fp32; the accumulation and the exact score's own chain as for bf16 (8 d 2^-24
//     covers d + 4 added terms with room to spare);
//   * the bias -c sig_i as three fp16 terms: residual <= 2^-33 |bias| or, where the third term would be subnormal, 2^-25;
//   * |x| > 65504 (a query row scaled by 1 / sig_u for DIRECT_MINUS_BOTH with sig_u below ~1e-5, an absurd embedding or c):
//     the operand is CLAMPED (no inf, hence no NaN, enters an accumulator) and its norm reported as +inf: an infinite
//     margin lists everything for that query (for every query when an item row did it), the lists overflow and the exact
//     kernel ranks -- right, slow, and not a case a trained model produces.
//   * RUBI_BOTH lists acc'' * sig_u: off by sig_u times the accumulator's margin -- the thresholds, the listing test and
//     k_select_b scale the margin by sig_u there (a query with sig_u = 0.01 has scores, and score gaps, a hundred times smaller).
// filter_rel_h(d) = 1.001 * 2^-10 + 8 d 2^-24: d = 32: 9.93e-4, 64: 1.008e-3, 128: 1.039e-3, 256: 1.100e-3.
// ----------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__host__ __device__ constexpr float filter_rel_h(int d) { return 1.001f / 1024.f + 8.f * (float)d / 16777216.f; }
constexpr float kHalfTiny = 3.0e-8f;         // > 2^-25: what rounding to fp16 costs an operand below fp16's normal range
constexpr float kHalfMax = 65504.f;
__device__ __forceinline__ float filter_margin_h(int d, float unorm, float qmax, float c) {
    const float b = unorm * qmax * 1.0001f;
    const float m = filter_rel_h(d) * b + kFilterAbs * (b + fabsf(c)) + kHalfTiny * (sqrtf((float)d) * (unorm + qmax) + 2.0f) + 1e-37f;
    return m == m ? m : INFINITY;            // (inf * 0: a flagged query against an all-zero catalogue)
}
// the margin of the filter in use: half != 0 = the fp16 filter
__device__ __forceinline__ float filter_margin_of(int half, int d, float unorm, float qmax, float c) {
    return half ? filter_margin_h(d, unorm, qmax, c) : filter_margin(d, unorm, qmax, c);
}
__device__ __forceinline__ uint32_t f16_rne_bits(float f) {            // round to nearest even, clamped to the finite range
    f = fminf(fmaxf(f, -kHalfMax), kHalfMax);
    const _Float16 hv = (_Float16)f;
    uint16_t b;
    __builtin_memcpy(&b, &hv, 2);
    return b;
}
__device__ __forceinline__ float f16_bits_value(uint32_t b) {
    const uint16_t s = (uint16_t)b;
    _Float16 hv;
    __builtin_memcpy(&hv, &s, 2);
    return (float)hv;
}

__device__ __forceinline__ uint32_t bf16_rne_bits(float f) {          // round to nearest even; NaN stays NaN
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// Rows of the item table (shard) and the query users' rows -> bf16 copies; |u| per query user; max |q| over the items
// (atomicMax on the float's bits: norms are >= 0).  One 8-lane group per 64 columns... one lane converts 8 columns.
constexpr int kPrepTrips = 4;         // (rows per lane group and block; measured on the Gowalla shape: 8 -> 4: k_bf16_prep 17 -> 13.5 us)
constexpr int kPrepTripsAlone = 2;    // k_bf16_prep_c by itself: 11 us; beside the seed blocks of k_prep_tau_seed 4 is better (32 vs 35 us)
template <int D>
__device__ __forceinline__ void bf16_prep_block(int blk, int U, int n_local, const float *__restrict__ users_tab,
                                                const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                uint4 *__restrict__ users_bf, uint4 *__restrict__ items_bf,
                                                float *__restrict__ unorm, uint32_t *__restrict__ qmax_bits) {
    constexpr int LPRB = D / 8, RPB = 256 / LPRB;            // lanes per row, rows per block and trip
    __shared__ float s_max[4];
    const int sub = threadIdx.x % LPRB, slot = threadIdx.x / LPRB;
    // kPrepTrips rows per lane group, all loads first: an eighth of the blocks -- and of the atomics on the one maximum,
    // which were most of this kernel's time with a block per 32 rows
    float4 a[kPrepTrips], b[kPrepTrips];
    const float *src[kPrepTrips];
    long long row[kPrepTrips];
#pragma unroll
    for (int t = 0; t < kPrepTrips; ++t) {
        row[t] = ((long long)blk * kPrepTrips + t) * RPB + slot;
        const bool is_item = row[t] < n_local;
        const long long q = row[t] - n_local;
        src[t] = is_item ? items + (size_t)row[t] * D : users_tab + (size_t)((q < U) ? (user_ids ? user_ids[q] : q) : 0) * D;
    }
#pragma unroll
    for (int t = 0; t < kPrepTrips; ++t) { a[t] = ld4(src[t] + 8 * sub); b[t] = ld4(src[t] + 8 * sub + 4); }
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < kPrepTrips; ++t) {
        const bool is_item = row[t] < n_local;
        const long long q = row[t] - n_local;
        float sq = 0.f;
        if (is_item || q < U) {
            const float x[8] = {a[t].x, a[t].y, a[t].z, a[t].w, b[t].x, b[t].y, b[t].z, b[t].w};
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hi[k] = bf16_rne_bits(x[k]);
                lo[k] = bf16_rne_bits(x[k] - __uint_as_float(hi[k] << 16));       // (x - hi is exact in fp32)
            }
            // row layout: hi[D] then lo[D]
            uint4 *dst = is_item ? items_bf + (size_t)row[t] * 2 * LPRB : users_bf + (size_t)q * 2 * LPRB;
            dst[sub] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
            dst[LPRB + sub] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
            sq = dot4(a[t], a[t]) + dot4(b[t], b[t]);
        }
        sq = group_sum<LPRB>(sq);
        float nrm = sqrtf(sq) * 1.0001f;                      // (rounded up: the margin must not be short)
        if (!(nrm == nrm)) nrm = INFINITY;                    // a NaN row: margin +inf, everything of that user is listed / re-scored
        if (!is_item && q < U && sub == 0) unorm[q] = nrm;
        if (is_item) m = fmaxf(m, nrm);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (m > __uint_as_float(__hip_atomic_load(qmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(qmax_bits, __float_as_uint(m));
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_bf16_prep(int U, int n_local, const float *__restrict__ users_tab,
                                                   const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                   uint4 *__restrict__ users_bf, uint4 *__restrict__ items_bf,
                                                   float *__restrict__ unorm, uint32_t *__restrict__ qmax_bits) {
    bf16_prep_block<D>(blockIdx.x, U, n_local, users_tab, user_ids, items, users_bf, items_bf, unorm, qmax_bits);
}
static inline unsigned bf16_prep_blocks(int U, int n_local, int d) {
    const size_t rows_per_block = (size_t)kPrepTrips * (256 / (d / 8));
    return (unsigned)(((size_t)n_local + U + rows_per_block - 1) / rows_per_block);
}

// Test-only (macr_test_bf16_products): the RAW product of the bf16 kernels for a whole (U, N) block -- the three MFMAs
// per 16 k of k_score_stream_b / k_score_sample_b / k_score_stream_bs in their order, on the operand copies k_bf16_prep
// wrote -- and the margin the filter would grant each query.  One wave per 32 x 32 tile.
template <int D>
__global__ __launch_bounds__(64) void k_test_bf16_products(int U, int N, const uint4 *__restrict__ users_bf,
                                                           const uint4 *__restrict__ items_bf, const float *__restrict__ unorm,
                                                           const uint32_t *__restrict__ qmax_bits, float c,
                                                           float *__restrict__ prod, float *__restrict__ margin) {
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    const int item = min((int)blockIdx.x * 32 + col, N - 1), user = min((int)blockIdx.y * 32 + col, U - 1);
    const uint4 *irow = items_bf + (size_t)item * (2 * D / 8), *urow = users_bf + (size_t)user * (2 * D / 8);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int sI = 0; sI < D / 16; ++sI) {
        uint4 v;
        v = irow[2 * sI + h];          const bf16x8 ah = *reinterpret_cast<bf16x8 *>(&v);
        v = irow[D / 8 + 2 * sI + h];  const bf16x8 al = *reinterpret_cast<bf16x8 *>(&v);
        v = urow[2 * sI + h];          const bf16x8 bh = *reinterpret_cast<bf16x8 *>(&v);
        v = urow[D / 8 + 2 * sI + h];  const bf16x8 bl = *reinterpret_cast<bf16x8 *>(&v);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    const int u = (int)blockIdx.y * 32 + col;
    if (u >= U) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int it = (int)blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (it < N) prod[(size_t)u * N + it] = acc[r];
    }
    if (blockIdx.x == 0 && h == 0) margin[u] = filter_margin(D, unorm[u], __uint_as_float(*qmax_bits), c);
}

template <int D>
struct StreamCfgB {
    static constexpr int RSB = 2 * D + 8;                     // LDS row (hi[D], lo[D]) stride in bf16: 4D + 16 bytes, conflict-free 16-byte reads
    static constexpr int NS = D / 16;                         // k slabs per tile (32x32x16), three MFMAs each
    static constexpr int UNITS = kTileItems * 2 * D / 8;      // 16-byte units per tile
    static constexpr size_t smem = (size_t)2 * kTileItems * RSB * 2 + 4 * kTileItems * 4 + kUsersPerBlock * 4 + 16;
};

// The listing test of k_score_stream_b in the domain of the RAW product: score(acc) >= tau  <=>  acc >= fma(X_u, Y_i, Z_u)
// (sigmoids > 0), one fma + one compare per score instead of the epilogue + compare; the epilogue runs for the hits only.
//   RUBI_BOTH  (acc - c) s_i s_u >= tau   X = tau / s_u   Y = 1 / s_i   Z = c
//   RUBI       (acc - c) s_i     >= tau   X = tau         Y = 1 / s_i   Z = c
//   DIRECT_MINUS       acc - c s_i     >= tau   X = c        Y = s_i    Z = tau
//   DIRECT_MINUS_BOTH  acc - c s_i s_u >= tau   X = c s_u    Y = s_i    Z = tau
// The roundings of X, Y and the fma are inside the slack the threshold already carries (filter_margin's absolute term).
// A sigmoid below 1e-30 (1/s overflows, tau/s is 0/0 for tau = 0) sends the tile / the wave through the plain epilogue.
// Masked items are not poisoned in the accumulator here (16 selects per tile and wave): the mask is tested for the hits.
template <int KIND>
__device__ __forceinline__ float filter_y(float sig) {
    return (KIND == MACR_SCORE_RUBI_BOTH || KIND == MACR_SCORE_RUBI) ? 1.0f / sig : sig;
}

// The listing pass of k_score_stream (MODE = list, one c, first round) on bf16 copies of the operands; see above.
// A wave owns UG groups of 32 users.  UG = 1 (8 waves per 256-user block, four waves per SIMD up to d = 64) is what runs:
// measured on the Gowalla shape, UG = 2 (4-wave blocks, every item fragment read from LDS feeding 6 MFMAs, 198 VGPRs, two
// waves per SIMD) took 362 us against 291 -- the pass is a chain of latencies (LDS reads, the barrier, the trip of the
// next tile), and waves in flight hide them better than longer MFMA runs do; two tiles per barrier (spills at 128 VGPRs)
// and a tile fetched two visits ahead (the register copies wait for the loads) were slower too.
template <int D>
struct StreamGroupsB { static constexpr int UG = 1, NW = kUsersPerBlock / (32 * UG), THREADS = 64 * NW; };

template <int D, int KIND, bool REPAIR = false>
__global__ __launch_bounds__(StreamGroupsB<D>::THREADS, D <= 64 ? 4 : 2) void k_score_stream_b(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items_bf,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits,
    const float *__restrict__ sig_u, const float *__restrict__ sig_i, float c_val, const float *__restrict__ c_dev,
    const uint32_t *__restrict__ mask_bits, int item_offset, int ublocks, const float *__restrict__ tau,
    uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow, int ovf_per_user,
    int32_t *blk_flag, const int32_t *__restrict__ ub_map, const int32_t *__restrict__ n_ub_dev, int slots_full) {
    using C = StreamCfgB<D>;
    constexpr int UG = StreamGroupsB<D>::UG, THREADS = StreamGroupsB<D>::THREADS;
    constexpr int LDU = (C::UNITS + THREADS - 1) / THREADS;
    constexpr int kCheckTiles = 8;
    const float c = c_dev ? *c_dev : c_val;
    constexpr int RSB = C::RSB, NS = C::NS;
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *s_a = reinterpret_cast<__bf16 *>(smem);                                   // [2][32][RSB]
    float *s_sig = reinterpret_cast<float *>(smem + (size_t)2 * kTileItems * RSB * 2);  // [2][32]
    float *s_y = s_sig + 2 * kTileItems;                                               // [2][32]  filter_y(sig_i)
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_y + 2 * kTileItems);            // [256]
    int *s_slow = reinterpret_cast<int *>(s_cnt + kUsersPerBlock);                    // [2]  tile holds a sigmoid < 1e-30

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    // repair round (REPAIR; k_score_stream): only the *n_ub_dev user blocks of ub_map, by part of the grid
    int n_ub = ublocks;
    long long G = gridDim.x;
    const long long b = blockIdx.x;
    RepairLayout rl = {false, 0, U};
    if (REPAIR) {
        n_ub = *n_ub_dev;
        if (n_ub == 0) return;
        rl = repair_layout(slots_full, U, n_ub);
        G = repair_grid(rl, G, ublocks, n_ub, T);
        if (b >= G) return;
    }
    int S = (int)(0.6180339f * (float)T);                     // the visit order of k_score_stream (scattered tile ranges)
    S = S < 1 ? 1 : S;
    for (;; ++S) {
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T); };
    auto visit_after = [&](int tile) { const int n = tile + S; return n >= T ? n - T : n; };     // visit(i + 1) from visit(i)
    const float qmax = __uint_as_float(*qmax_bits);
    const long long W = (long long)n_ub * T;
    const long long w_end = W * (b + 1) / G;
    for (long long w = W * b / G; w < w_end;) {
    const int ubv = (int)(w / T), i0 = (int)(w - (long long)ubv * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ubv * T * G / W;
    while (W * (first + 1) / G <= (long long)ubv * T) ++first;
    while (W * first / G > (long long)ubv * T) --first;
    const int split = (int)(b - first);
    const int ub = REPAIR ? ub_map[ubv] : ubv;

    for (int k = tid; k < kUsersPerBlock; k += THREADS) s_cnt[k] = 0u;
    // per user group g: users (wid * UG + g) * 32 + col
    int uslot[UG], q[UG];
    bool q_ok[UG];
    bf16x8 bhi[UG][NS], blo[UG][NS];
    float su[UG], tau_s[UG], fx[UG], fz[UG];
    uint64_t *my_list[UG];
    bool wave_slow = false;
#pragma unroll
    for (int g = 0; g < UG; ++g) {
        uslot[g] = (wid * UG + g) * 32 + col;
        q[g] = ub * kUsersPerBlock + uslot[g];
        q_ok[g] = q[g] < U;
        const uint4 *urow = users_bf + (size_t)(q_ok[g] ? q[g] : 0) * (2 * D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
            if (!q_ok[g]) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[g][sI] = *reinterpret_cast<bf16x8 *>(&v);
            blo[g][sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
        su[g] = (score_uses_sig_u(KIND) && q_ok[g]) ? sig_u[q[g]] : 1.0f;
        // listing test: score_bf16 >= tau - margin (NaN = never: padding users)
        tau_s[g] = __builtin_nanf("");
        if (q_ok[g]) tau_s[g] = tau[q[g]] - 1.01f * filter_margin(D, unorm[q[g]], qmax, c);
        // the test on the raw product (filter_y): acc >= fma(fx, Y_i, fz)
        fx[g] = 0.f; fz[g] = tau_s[g];
        if (KIND == MACR_SCORE_RUBI_BOTH) { fx[g] = tau_s[g] / su[g]; fz[g] = c; }
        else if (KIND == MACR_SCORE_RUBI) { fx[g] = tau_s[g]; fz[g] = c; }
        else if (KIND == MACR_SCORE_DIRECT_MINUS) { fx[g] = c; }
        else if (KIND == MACR_SCORE_DIRECT_MINUS_BOTH) { fx[g] = c * su[g]; }
        if (KIND == MACR_SCORE_RUBI_BOTH) wave_slow = wave_slow || __any(q_ok[g] && !(su[g] > 1e-30f));
        my_list[g] = lists + ((size_t)split * rl.stride + ((REPAIR && rl.compact) ? ubv * kUsersPerBlock + uslot[g] : (q_ok[g] ? q[g] : 0))) * cap;
    }

    int vi = i0, t = visit(i0);
    uint4 stg[LDU];
    float sg = 0.f;
    uint32_t tm_next[UG];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            const int e = tid + THREADS * k, row = (e / (2 * D / 8)) & (kTileItems - 1), c8 = e % (2 * D / 8);
            const int it = min(tile * kTileItems + row, n_local - 1);
            stg[k] = items_bf[(size_t)it * (2 * D / 8) + c8];
        }
        if (score_uses_sig_i(KIND)) sg = sig_i[min(tile * kTileItems + (tid & (kTileItems - 1)), n_local - 1)];
#pragma unroll
        for (int g = 0; g < UG; ++g) tm_next[g] = (mask_bits && q_ok[g]) ? mask_bits[(size_t)tile * U + q[g]] : 0u;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            const int e = tid + THREADS * k, row = e / (2 * D / 8), c8 = e % (2 * D / 8);
            if (row < kTileItems)
                *reinterpret_cast<uint4 *>(s_a + ((size_t)buf * kTileItems + row) * RSB + 8 * c8) = stg[k];
        }
        if (score_uses_sig_i(KIND)) {
            asm volatile("" : "+v"(sg));
            if (tid < 64) {                                   // (wave 0; lanes 32-63 hold copies of the tile's 32 values)
                const bool tiny = !(sg > 1e-30f);
                if (tid < kTileItems) { s_sig[buf * kTileItems + tid] = sg; s_y[buf * kTileItems + tid] = filter_y<KIND>(sg); }
                const bool any_tiny = __any(tiny);
                if (tid == 0) s_slow[buf] = any_tiny ? 1 : 0;
            }
        }
    };

    int buf = 0;
    if (vi < i1) { load_tile(t); store_tile(0); }
    uint32_t tm_cur[UG];
#pragma unroll
    for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
    __syncthreads();
    while (vi < i1) {
        const bool has_next = vi + 1 < i1;
        const int tn = has_next ? visit_after(t) : t;
        if (has_next) load_tile(tn);

        const int gid0 = t * kTileItems + item_offset;
        const int valid = n_local - t * kTileItems;           // < 32 only in the last tile of the shard
        const uint32_t tail = valid < kTileItems ? (valid > 0 ? ~0u << valid : ~0u) : 0u;

        const __bf16 *ua = s_a + ((size_t)buf * kTileItems + col) * RSB + 8 * h;
        f32x16 acc[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
#pragma unroll
            for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[g][sI], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[g][sI], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[g][sI], acc[g], 0, 0, 0);
        }

        const bool slow = score_uses_sig_i(KIND) && (wave_slow || s_slow[buf] != 0);       // wave-uniform, practically never
        float yi[16];
        if (score_uses_sig_i(KIND)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<float4 *>(yi + 4 * k) = *reinterpret_cast<const float4 *>(s_y + buf * kTileItems + 8 * k + 4 * h);
        }
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            uint32_t hit = 0u;                                // bit r: some lane's score r passes (scalar unit)
            if (!slow) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    hit |= __ballot(acc[g][r] >= (score_uses_sig_i(KIND) ? fmaf(fx[g], yi[r], fz[g]) : fz[g])) ? 1u << r : 0u;
            } else {
                hit = 0xffffu;                                // every score through the epilogue below
            }
            if (hit) {                                        // one wave-uniform branch per tile and group
                const uint32_t tmask = tm_cur[g] | tail;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!((hit >> r) & 1u)) continue;         // wave-uniform: nobody's score r passes
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    bool pass = slow || acc[g][r] >= (score_uses_sig_i(KIND) ? fmaf(fx[g], yi[r], fz[g]) : fz[g]);
                    pass = pass && ((tmask >> row) & 1u) == 0u;   // masked (train) items, rows past the end of the shard
                    if (pass) {
                        float v = acc[g][r];
                        if (score_uses_sig_i(KIND)) v = score_epilogue<KIND>(v, c, s_sig[buf * kTileItems + row], su[g]);
                        if (v >= tau_s[g]) {
                            const uint32_t pos = atomicAdd(&s_cnt[uslot[g]], 1u);
                            if (pos < (uint32_t)cap) my_list[g][pos] = make_key(v, gid0 + row);
                            else {                            // full: stop listing for her
                                overflow[ovf_per_user ? q[g] : 0] = 1;
                                tau_s[g] = INFINITY; fz[g] = INFINITY;
                                if (KIND == MACR_SCORE_RUBI_BOTH || KIND == MACR_SCORE_RUBI) fx[g] = INFINITY;
                            }
                        }
                    }
                }
            }
        }
        bool stop = false;                                    // block-uniform; see k_score_stream (stale seeds)
        if (blk_flag) {
            const int done = vi - i0 + 1;
            const bool check = done == 2 || done == kCheckTiles || done == 4 * kCheckTiles;
            if (check || (done & 7) == 0) {
                bool mine = false;
                if (check) {
                    uint32_t a = 0u;                          // appended so far by this wave's 32 UG users
                    if (h == 0) {
#pragma unroll
                        for (int g = 0; g < UG; ++g) a += s_cnt[uslot[g]];
                    }
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, kWave);
                    const float usable = 32.f * UG * (float)(done * kTileItems) * (float)(kSelRegs * 64) / (float)n_local;
                    mine = (float)a > usable + 64.f * UG;
                } else if (tid == 0) {
                    mine = __hip_atomic_load(blk_flag + ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                }
                stop = __syncthreads_or(mine) != 0;
                if (stop && tid == 0) __hip_atomic_store(blk_flag + ub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
        buf ^= 1;
        t = tn; ++vi;
        if (stop) break;
    }
    for (int k = tid; k < kUsersPerBlock; k += THREADS) {
        const int qq = ub * kUsersPerBlock + k;
        const int ql2 = (REPAIR && rl.compact) ? ubv * kUsersPerBlock + k : qq;
        if (qq < U) counts[(size_t)split * rl.stride + ql2] = (int32_t)min(s_cnt[k], (uint32_t)cap);
    }
    }   // segments
}

// ============================================================================
// The listing pass with the epilogue INSIDE the matrix product (k_score_stream_c; what macr_score_topk runs under the bf16
// filter -- k_score_stream_b above stays for reference and its bound test, k_score_stream_bs for the c sweep).
//
// A stand-alone bench of this kernel and its predecessors (round 4; cycle-counter traces, PMC: profiles/r04_listing_*) says what a visit of
// k_score_stream_b costs a wave: a third of it are the MFMAs; the rest is vector-ALU work that four waves per SIMD queue
// up for -- the address arithmetic of the tile copy (per-thread divisions and clamps), an fma + compare + three scalar
// instructions per score for the test on the raw product, the staging of sig_i and 1/sig_i -- and 12 instead of 9 LDS
// fragment reads.  Here:
//   * the operand copies carry the epilogue.  Items: q'_i = s_i q_i (s_i = sig_i for the RUBI kinds, 1 otherwise), and one
//     more 16-byte unit per row with the three-term bf16 split of the bias b_i = -c sig_i (0 for NORMAL), multiplied in the
//     LAST MFMA against a user-side slab of ones.  Users: u' = u / sig_u for DIRECT_MINUS_BOTH, u otherwise.  Then
//         acc'' = u'.q' + b_i      listed score v = acc'' * f_u      (f_u = sig_u for the *_BOTH kinds, 1 otherwise)
//     is the score of every kind, the listing test is acc'' >= (tau - margin) / f_u -- ONE compare against a lane constant --
//     and nothing per item is staged beside the tile.  (The products accumulate at their own magnitude and the bias joins
//     last: one rounding at the magnitude of c, as in the fp32 epilogue's (y - c).  A user with sig_u < 1e-30 keeps u' = u
//     and a zero in place of the ones: her DIRECT_MINUS_BOTH score is y to 1e-30 |c|.)
//     |v_bf16 - v_fp32| <= filter_margin as before: s_i, sig_u <= 1 only shrink the products the bound is stated on, the
//     scaled operands and the bias split add roundings of 2^-24 relative (inside the margin's absolute term).
//   * item rows in global memory exactly as they sit in LDS: RU = 2D/8 + 1 units of 16 bytes (hi[D], lo[D], bias unit; the
//     odd unit stride that makes the fragment reads conflict-free), the table padded to whole tiles, so a tile is TU
//     consecutive units: thread t copies unit t (+ 512 k), the tile offset is wave-uniform (SGPRs), nothing is clamped.
//     Every load is unconditional (a load under a branch makes the compiler wait for it at the join).
//   * the visit loop is unrolled over the two LDS buffers (immediate offsets); all nine fragment reads first, then the 13
//     MFMAs back to back; per score one v_cmp + one s_cbranch; a hit costs eight vector instructions (mask bit, LDS counter,
//     key, store).  A full list keeps counting (the segment's end flags it) instead of raising its threshold.
// Measured (that bench, Gowalla shape, ~176 listed per query): 317 -> 246-252 us; instructions per MFMA:
// VALU 6.7 -> 3.2, SALU 8.4 -> 2.1; the matrix pipe 40 % -> 52 % busy at the clock the chip then sustains (1.95 GHz).
// ============================================================================
template <int D>
struct StreamCfgC {
    static constexpr int RU = 2 * D / 8 + 1;                  // 16-byte units per item row
    static constexpr int TU = kTileItems * RU;                // ... per tile
    static constexpr int NS = D / 16;
    static constexpr int LDU = (TU + 511) / 512, REM = TU - 512 * (LDU - 1);      // copy rounds of 512 threads; units of the last one
    static constexpr size_t smem = (size_t)2 * TU * 16 + kUsersPerBlock * 4 + 16;
};
static inline size_t items_c_bytes(int n_local, int d) { return (size_t)n_tiles(n_local) * kTileItems * (2 * d / 8 + 1) * 16; }
// the fp16 filter's copies (bf16_prep_c_block<.., HALF>): RU = D/8 + 1 units per item row; they live in the same buffers
template <int D>
struct StreamCfgH1 {                                           // one tile per visit (k_score_sample_c<.., HALF>)
    static constexpr int RU = D / 8 + 1;
    static constexpr int TU = kTileItems * RU;
    static constexpr int NS = D / 16;
    static constexpr int LDU = (TU + 511) / 512, REM = TU - 512 * (LDU - 1);
    static constexpr size_t smem = (size_t)2 * TU * 16 + kUsersPerBlock * 4 + 16;
};
// Tiles per step of k_score_stream_h (= 4-wave groups of its block).  1 (default): a block is FOUR waves, 256 queries x one tile
// per step, four blocks per CU; 2: eight waves, the two halves on two tiles of one step, two blocks per CU.  A step is a latency
// chain (load -> wait -> ds_write -> barrier -> fragment reads -> wait) and four independent chains per CU hide it better than two:
// listing pass 167 -> 134 us sampled, 145 -> 123 seeded on the Gowalla shape (-DMACR_H_TPS=2 for the A/B; profiles/r06_eval_f16_filter.txt).
// The price: twice the blocks, shorter tile ranges, 18 result slots per query instead of 10 (k_select_b +2-4 us).
#ifndef MACR_H_TPS
#define MACR_H_TPS 1
#endif
constexpr int kHTiles = MACR_H_TPS;
template <int D>
struct StreamCfgH {                                            // kHTiles tiles per step (k_score_stream_h)
    static constexpr int RU = D / 8 + 1;
    static constexpr int TU = kTileItems * RU;
    static constexpr int NS = D / 16;
    static constexpr int THREADS = 256 * kHTiles;
    static constexpr int SU = kHTiles * TU;                    // units staged per step
    static constexpr int LDU = (SU + THREADS - 1) / THREADS;
    static constexpr size_t smem = (size_t)2 * SU * 16 + kUsersPerBlock * 4 + 16;
};
// eight fp16 numbers per operand register group, carried in the bf16 vector type of the surrounding code
__device__ __forceinline__ f32x16 mfma_f16(bf16x8 a, bf16x8 b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

// The bias slab of the fp16 filter.  DIRECT_MINUS_BOTH subtracts c sig_i sig_u: the bf16 copies divide the query row by sig_u
// and multiply the listed score by it (a bf16 slab entry holds 8 bits of sig_u); 1 / sig_u can leave fp16's range, so here both
// factors stay in the slab instead, as three fp16 terms each -- item side b = -c sig_i ~ b1 + b2 + b3, query side
// s = sig_u ~ s1 + s2 + s3 -- and six of the nine cross products ride in the slab's k slots:
//   item unit  [b1 b1 b2 b1 b2 b3 0 0],  query unit [s1 s2 s1 s3 s2 s1 0 0]      (dropped: b2 s3 + b3 s2 + b3 s3 <= 2^-32 |b s|)
// every product exact in fp32.  The other kinds put s = 1 there (s1 = 1, s2 = s3 = 0: the slab adds b1 + b2 + b3).
// |b s - slab| <= 2^-25 (|c| + 1) where terms fall below fp16's subnormals: inside filter_margin_h's absolute terms.
template <int KIND> __device__ __forceinline__ bf16x8 bias_query_unit_h(float su, bool on) {
    const float sv = KIND == MACR_SCORE_DIRECT_MINUS_BOTH ? su : 1.0f;
    const uint32_t s1 = f16_rne_bits(sv);
    const float r1 = sv - f16_bits_value(s1);
    const uint32_t s2 = f16_rne_bits(r1), s3 = f16_rne_bits(r1 - f16_bits_value(s2));
    union { uint32_t u[4]; bf16x8 v; } w;
    w.u[0] = on ? (s1 | (s2 << 16)) : 0u; w.u[1] = on ? (s1 | (s3 << 16)) : 0u; w.u[2] = on ? (s2 | (s1 << 16)) : 0u; w.u[3] = 0u;
    return w.v;
}
__device__ __forceinline__ uint4 bias_item_unit_h(float bias) {
    const float bc = fminf(fmaxf(bias, -kHalfMax), kHalfMax);
    const uint32_t b1 = f16_rne_bits(bc);
    const float r1 = bc - f16_bits_value(b1);
    const uint32_t b2 = f16_rne_bits(r1), b3 = f16_rne_bits(r1 - f16_bits_value(b2));
    return make_uint4(b1 | (b1 << 16), b2 | (b1 << 16), b2 | (b3 << 16), 0u);
}
// the factor a listed score carries outside the accumulator under the fp16 filter
template <int KIND> __device__ __forceinline__ float query_factor_h(float su) { return KIND == MACR_SCORE_RUBI_BOTH ? su : 1.0f; }

// item scale / bias / user scale of a score kind (see above); c < 0 or > 0 alike
template <int KIND> __device__ __forceinline__ float item_scale_c(float sgi) {
    return (KIND == MACR_SCORE_RUBI_BOTH || KIND == MACR_SCORE_RUBI) ? sgi : 1.0f;
}
template <int KIND> __device__ __forceinline__ float item_bias_c(float sgi, float c) {
    return KIND == MACR_SCORE_NORMAL ? 0.0f : -(c * sgi);
}
__device__ __forceinline__ bool sig_u_tiny(float su) { return !(su > 1e-30f); }

// Operand copies for k_score_stream_c: item rows (and the zero rows up to a whole tile) in the RU-unit layout, query rows
// as in k_bf16_prep (scaled for DIRECT_MINUS_BOTH), |u| per query, max |q| -- norms of the UNscaled rows, which bound the
// scaled ones.  Row space of a block: [0, n_pad) items, [n_pad, n_pad + U) queries.
// HALF: the fp16 filter's copies instead -- fp16[D] per row (item rows: + the bias unit, RU = D/8 + 1), clamped to the
// finite range; a row that was clamped reports an infinite norm (filter_margin_h).
// FUSED (k_eval_prologue_prep): the branch factors are not read but COMPUTED here, from the rows this block holds anyway --
// sigmoid(row . w), k_branch_sigmoid's arithmetic bit for bit (a lane's two float4 are two of its lanes; its butterfly pairs
// them last) -- and written to sig_i_out / sig_u_out; the block's largest item norm goes to qpart[blk] instead of an atomic
// max (the launch also zeroes the workspace head: no word of it may be accumulated into), for qmax_from_parts to reduce.
template <int D, int KIND, int TRIPS, bool HALF = false, bool FUSED = false>
__device__ __forceinline__ void bf16_prep_c_block(int blk, int U, int n_local, const float *__restrict__ users_tab,
                                                  const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                  const float *__restrict__ sig_u, const float *__restrict__ sig_i, float c,
                                                  uint4 *__restrict__ users_c, uint4 *__restrict__ items_c,
                                                  float *__restrict__ unorm, uint32_t *__restrict__ qmax_bits,
                                                  const float *__restrict__ w_item = nullptr, const float *__restrict__ w_user = nullptr,
                                                  float *__restrict__ sig_i_out = nullptr, float *__restrict__ sig_u_out = nullptr,
                                                  float *__restrict__ qpart = nullptr) {
    constexpr int LPRB = D / 8, RPB = 256 / LPRB, RU = HALF ? D / 8 + 1 : StreamCfgC<D>::RU;
    __shared__ float s_max[4];
    const int n_pad = ((n_local + kTileItems - 1) / kTileItems) * kTileItems;
    const int sub = threadIdx.x % LPRB, slot = threadIdx.x / LPRB;
    float4 a[TRIPS], b[TRIPS];
    float scale[TRIPS], bias[TRIPS];
    long long row[TRIPS];
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        row[t] = ((long long)blk * TRIPS + t) * RPB + slot;
        const bool is_item = row[t] < n_local;
        const long long q = row[t] - n_pad;
        const bool is_user = q >= 0 && q < U;
        const float *src = is_item ? items + (size_t)row[t] * D : users_tab + (size_t)(is_user ? (user_ids ? user_ids[q] : q) : 0) * D;
        a[t] = ld4(src + 8 * sub); b[t] = ld4(src + 8 * sub + 4);
        scale[t] = 1.0f; bias[t] = 0.0f;
        float sg_row = 0.0f;
        if (FUSED) {
            const float *wv = is_item ? w_item : w_user;      // (uniform over the row's lanes; w_user == NULL: one-branch scores)
            float pa = 0.f, pb = 0.f;
            if (wv && (is_item || is_user)) { pa = dot4(a[t], ld4(wv + 8 * sub)); pb = dot4(b[t], ld4(wv + 8 * sub + 4)); }
#pragma unroll
            for (int m = LPRB >> 1; m >= 1; m >>= 1) { pa += __shfl_xor(pa, m, kWave); pb += __shfl_xor(pb, m, kWave); }
            sg_row = sigmoid_acc(pa + pb);
            if (sub == 0 && is_item && sig_i_out) sig_i_out[row[t]] = sg_row;
            if (sub == 0 && is_user && w_user && sig_u_out) sig_u_out[q] = sg_row;
        }
        if (is_item && score_uses_sig_i(KIND)) {
            const float sgi = FUSED ? sg_row : sig_i[row[t]];
            scale[t] = item_scale_c<KIND>(sgi); bias[t] = item_bias_c<KIND>(sgi, c);
        }
        if (is_user && KIND == MACR_SCORE_DIRECT_MINUS_BOTH && !HALF) {     // (HALF: sig_u sits in the bias slab, bias_query_unit_h)
            const float su = FUSED ? sg_row : sig_u[q];
            scale[t] = sig_u_tiny(su) ? 1.0f : 1.0f / su;
        }
    }
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        const bool is_item = row[t] < n_local, is_pad = row[t] >= n_local && row[t] < n_pad;
        const long long q = row[t] - n_pad;
        const bool is_user = q >= 0 && q < U;
        float sq = 0.f, clamped = 0.f;
        if (is_item || is_user || is_pad) {
            const float x[8] = {a[t].x * scale[t], a[t].y * scale[t], a[t].z * scale[t], a[t].w * scale[t],
                                b[t].x * scale[t], b[t].y * scale[t], b[t].z * scale[t], b[t].w * scale[t]};
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (HALF) {
                    hi[k] = is_pad ? 0u : f16_rne_bits(x[k]);
                    lo[k] = 0u;
                    if (!is_pad && fabsf(x[k]) > kHalfMax) clamped = 1.0f;
                } else {
                    hi[k] = is_pad ? 0u : bf16_rne_bits(x[k]);
                    lo[k] = is_pad ? 0u : bf16_rne_bits(x[k] - __uint_as_float(hi[k] << 16));
                }
            }
            const uint4 vh = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
            const uint4 vl = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
            if (is_user) {
                uint4 *dst = users_c + (size_t)q * (HALF ? 1 : 2) * LPRB;
                dst[sub] = vh;
                if (!HALF) dst[LPRB + sub] = vl;
            } else {
                uint4 *dst = items_c + (size_t)row[t] * RU;
                dst[sub] = vh;
                if (!HALF) dst[LPRB + sub] = vl;
                if (sub == 0) {                                // the bias unit: three bf16 (fp16) terms, then zeros
                    const float bi = is_pad ? 0.0f : bias[t];
                    if (HALF) {
                        if (fabsf(bi) > kHalfMax) clamped = 1.0f;
                        dst[LPRB] = bias_item_unit_h(bi);
                    } else {
                        const uint32_t b1 = bf16_rne_bits(bi);
                        const float r1 = bi - __uint_as_float(b1 << 16);
                        const uint32_t b2 = bf16_rne_bits(r1);
                        const uint32_t b3 = bf16_rne_bits(r1 - __uint_as_float(b2 << 16));
                        dst[2 * LPRB] = make_uint4(b1 | (b2 << 16), b3, 0u, 0u);
                    }
                }
            }
            sq = dot4(a[t], a[t]) + dot4(b[t], b[t]);         // (of the row as given, not as scaled)
        }
        sq = group_sum<LPRB>(sq);
        float nrm = sqrtf(sq) * 1.0001f;
        if (!(nrm == nrm)) nrm = INFINITY;
        if (HALF && group_sum<LPRB>(clamped) > 0.f) nrm = INFINITY;     // an operand outside fp16's range: no bound for this row
        if (is_user && sub == 0) unorm[q] = nrm;
        if (is_item) m = fmaxf(m, nrm);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (FUSED) qpart[blk] = m;
        else if (m > __uint_as_float(__hip_atomic_load(qmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(qmax_bits, __float_as_uint(m));
    }
}
// max |q| from the per-block maxima k_eval_prologue_prep left: one block of a later launch (k_tau_seed, k_score_sample_c)
__device__ __forceinline__ void qmax_from_parts(const float *__restrict__ qpart, int n_part, uint32_t *__restrict__ qmax_bits) {
    __shared__ float s_qm[16];
    float m = 0.f;
    for (int k = threadIdx.x; k < n_part; k += blockDim.x) m = fmaxf(m, qpart[k]);      // (norms: >= 0, +inf for a row without a bound)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & 63) == 0) s_qm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x + 63) / 64; ++k) m = fmaxf(m, s_qm[k]);
        *qmax_bits = __float_as_uint(m);
    }
    __syncthreads();
}
static inline unsigned bf16_prep_c_blocks(int U, int n_local, int d, int trips) {
    const size_t rows_per_block = (size_t)trips * (256 / (d / 8));
    return (unsigned)(((size_t)n_tiles(n_local) * kTileItems + U + rows_per_block - 1) / rows_per_block);
}

template <int D, int KIND, bool HALF = false>
__global__ __launch_bounds__(256) void k_bf16_prep_c(int U, int n_local, const float *__restrict__ users_tab,
                                                     const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                     const float *__restrict__ sig_u, const float *__restrict__ sig_i,
                                                     float c_val, const float *__restrict__ c_dev,
                                                     uint4 *__restrict__ users_c, uint4 *__restrict__ items_c,
                                                     float *__restrict__ unorm, uint32_t *__restrict__ qmax_bits) {
    bf16_prep_c_block<D, KIND, kPrepTripsAlone, HALF>(blockIdx.x, U, n_local, users_tab, user_ids, items, sig_u, sig_i, c_dev ? *c_dev : c_val, users_c,
                                                items_c, unorm, qmax_bits);
}

template <int D, int KIND, bool REPAIR = false>
__global__ __launch_bounds__(512, D <= 64 ? 4 : 2) void k_score_stream_c(
    int U, int n_local, const uint4 *__restrict__ users_c, const uint4 *__restrict__ items_c,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits,
    const float *__restrict__ sig_u, float c_val, const float *__restrict__ c_dev,
    const uint32_t *__restrict__ mask_bits, const uint32_t *__restrict__ zero_word, int item_offset, int ublocks,
    const float *__restrict__ tau, uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow,
    int ovf_per_user, int32_t *blk_flag, const int32_t *__restrict__ ub_map, const int32_t *__restrict__ n_ub_dev, int slots_full) {
    using C = StreamCfgC<D>;
    constexpr int THREADS = 512, NS = C::NS, RU = C::RU, TU = C::TU, LDU = C::LDU, REM = C::REM;
    constexpr int kCheckTiles = 8;
    const float c = c_dev ? *c_dev : c_val;
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *s_t = reinterpret_cast<uint4 *>(smem);                                      // [2][TU]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)2 * TU * 16);       // [256]
    int *s_stop = reinterpret_cast<int *>(s_cnt + kUsersPerBlock);                     // [1]  somebody wants the block to stop listing
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    int n_ub = ublocks;
    long long G = gridDim.x;
    const long long b = blockIdx.x;
    RepairLayout rl = {false, 0, U};
    if (REPAIR) {                                             // only the *n_ub_dev user blocks of ub_map, by part of the grid (k_score_stream)
        n_ub = *n_ub_dev;
        if (n_ub == 0) return;
        rl = repair_layout(slots_full, U, n_ub);
        G = repair_grid(rl, G, ublocks, n_ub, T);
        if (b >= G) return;
    }
    int S = (int)(0.6180339f * (float)T);                     // the visit order of k_score_stream (scattered tile ranges)
    S = S < 1 ? 1 : S;
    for (;; ++S) {
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T); };
    auto visit_after = [&](int tile) { const int n = tile + S; return n >= T ? n - T : n; };
    const float qmax = __uint_as_float(*qmax_bits);
    const long long W = (long long)n_ub * T;
    const long long w_end = W * (b + 1) / G;
    const uint4 *my_src = items_c + tid;                      // + tile * TU (wave-uniform)
    const int last_off = 512 * (LDU - 1) + (tid < REM ? 0 : tid % REM - tid);      // (everybody loads in the last round too: wrapped)
    const size_t mask_stride = mask_bits ? (size_t)U : 0;
    for (long long w = W * b / G; w < w_end;) {
    const int ubv = (int)(w / T), i0 = (int)(w - (long long)ubv * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ubv * T * G / W;
    while (W * (first + 1) / G <= (long long)ubv * T) ++first;
    while (W * first / G > (long long)ubv * T) --first;
    const int split = (int)(b - first);
    const int ub = REPAIR ? ub_map[ubv] : ubv;
    __syncthreads();                                          // (the previous segment's readers are done with s_t and s_cnt)
    for (int k = tid; k < kUsersPerBlock; k += THREADS) s_cnt[k] = 0u;
    if (tid == 0) *s_stop = 0;
    const int uslot = wid * 32 + col, q = ub * kUsersPerBlock + uslot;
    const bool q_ok = q < U;
    bf16x8 bhi[NS], blo[NS];
    {
        const uint4 *urow = users_c + (size_t)(q_ok ? q : 0) * (2 * D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
            if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[sI] = *reinterpret_cast<bf16x8 *>(&v);
            blo[sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
    }
    const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;
    const bool tiny = score_uses_sig_u(KIND) && sig_u_tiny(su);
    // listed score = acc'' * fu
    const float fu = (KIND == MACR_SCORE_DIRECT_MINUS_BOTH && tiny) ? 1.0f : su;
    // the user side of the bias slab: ones against the three bias terms (k = 0, 1, 2; lanes 0-31 hold k < 8), zeros elsewhere
    const bool ones_on = KIND != MACR_SCORE_NORMAL && h == 0 && !(KIND == MACR_SCORE_DIRECT_MINUS_BOTH && tiny);
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = ones_on ? 0x3f803f80u : 0u; ones.u[1] = ones_on ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    const bf16x8 bext = ones.v;
    // listing test: v_bf16 >= tau - margin  <=>  acc'' >= (tau - margin) / fu   (fu > 0; NaN = never: padding queries)
    float tau_s = __builtin_nanf("");
    if (q_ok) tau_s = tau[q] - 1.01f * filter_margin(D, unorm[q], qmax, c);
    float thr = tau_s / fu;
    if (KIND == MACR_SCORE_RUBI_BOTH && tiny) thr = q_ok ? -INFINITY : thr;       // (scores of the order of sig_u: everything is a candidate)
    uint64_t *my_list = lists + ((size_t)split * rl.stride + ((REPAIR && rl.compact) ? ubv * kUsersPerBlock + uslot : (q_ok ? q : 0))) * cap;
    const uint32_t *my_mask = mask_bits ? mask_bits + (q_ok ? q : 0) : zero_word;       // + tile * mask_stride (wave-uniform)
    uint32_t *my_cnt = &s_cnt[uslot];
    const int id_lane = item_offset + 4 * h;                  // id of accumulator slot r: tile * 32 + (r & 3) + 8 * (r >> 2) + id_lane

    uint4 stg[LDU];
    uint32_t tm_next = 0u;
    auto load_tile = [&](int tile) {
        const int tu = __builtin_amdgcn_readfirstlane(tile);
        const uint4 *src = my_src + (size_t)tu * TU;
#pragma unroll
        for (int k = 0; k + 1 < LDU; ++k) stg[k] = src[512 * k];
        stg[LDU - 1] = src[last_off];
        tm_next = my_mask[(size_t)tu * mask_stride];
    };
    auto store_tile = [&](auto BUF) {
        constexpr int buf = decltype(BUF)::value;
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            if (k + 1 < LDU || tid < REM) s_t[buf * TU + tid + 512 * k] = stg[k];
        }
    };
    // one visit: tile t sits in buffer BUF, the next tile's copy is in flight in stg
    auto one_visit = [&](auto BUF, int t, uint32_t tm_cur) {
        constexpr int buf = decltype(BUF)::value;
        const __bf16 *ua = reinterpret_cast<const __bf16 *>(s_t + buf * TU) + (size_t)col * (8 * RU) + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // fragments of up to four k-slabs (and the bias unit) in registers first, then their MFMAs back to back: an
        // instruction between two MFMAs on one accumulator costs ~43 cycles (MI355X_MICROARCH.md)
        constexpr int CH = NS < 4 ? NS : 4;
        bf16x8 ae;
        if (KIND != MACR_SCORE_NORMAL)
            ae = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(s_t + buf * TU) + (size_t)col * (8 * RU) + 2 * D);
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += CH) {
            bf16x8 ah[CH], al[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                ah[j] = *reinterpret_cast<const bf16x8 *>(ua + 16 * (s0 + j));
                al[j] = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * (s0 + j));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[j], bhi[s0 + j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[j], blo[s0 + j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[j], bhi[s0 + j], acc, 0, 0, 0);
            }
            if (KIND != MACR_SCORE_NORMAL && s0 + CH >= NS) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int valid = n_local - t * kTileItems;           // < 32 only in the last tile of the shard
        const uint32_t tailm = valid < kTileItems ? ~0u << (valid > 0 ? valid : 0) : 0u;
        const uint32_t tmh = (tm_cur | tailm) >> (4 * h);      // bit (r & 3) + 8 (r >> 2): this lane's row of slot r is masked / past the end
        const int id0 = t * kTileItems + id_lane;
        // (a row quad is tested together first -- v_max3 + v_max, one compare + branch -- and looked at score by score only where
        // some lane passes: k_score_stream_h's epilogue, measured there; 2/3 of the quads under sampled thresholds, 1/6 under seeded)
#pragma unroll
        for (int r4 = 0; r4 < 16; r4 += 4) {
            const float quad = fmaxf(fmaxf(fmaxf(acc[r4], acc[r4 + 1]), acc[r4 + 2]), acc[r4 + 3]);
            if (!__builtin_amdgcn_ballot_w64(quad >= thr)) continue;
#pragma unroll
            for (int r = r4; r < r4 + 4; ++r) {
                const bool above = acc[r] >= thr;
                if (__builtin_amdgcn_ballot_w64(above)) {      // wave-uniform: some lane's score r passes
                    const int rbit = (r & 3) + 8 * (r >> 2);
                    uint32_t m = tmh;
                    asm volatile("" : "+v"(m));                // (the mask test belongs in here, not in front of the branch)
                    if (above && !((m >> rbit) & 1u)) {
                        const uint32_t pos = atomicAdd(my_cnt, 1u);    // (keeps counting past cap: the segment's end flags it)
                        if (pos < (uint32_t)cap) my_list[pos] = make_key(acc[r] * fu, id0 + rbit);
                    }
                }
            }
        }
    };
    // Give up on stale seeds / follow the sibling blocks that did (k_score_stream).  A wave that wants the block to stop
    // raises s_stop; everybody reads it behind the visit's own barrier (a __syncthreads_or here cost three more barriers
    // per poll).  Returns whether this visit looks at the flag at all (block-uniform).
    int polled = 0;
    auto stale_mark = [&](int done) -> bool {
        if (!blk_flag) return false;
        const bool check = done == 2 || done == kCheckTiles || done == 4 * kCheckTiles;
        if (!(check || (done & 7) == 0)) return false;
        bool mine = false;
        if (check) {
            uint32_t a = h == 0 ? *my_cnt : 0u;                // appended so far by this wave's 32 users
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, kWave);
            const float usable = 32.f * (float)(done * kTileItems) * (float)(kSelRegs * 64) / (float)n_local;
            mine = (float)a > usable + 64.f;
        } else if (tid == 0) {
            // the siblings' mark: what the PREVIOUS poll fetched (the load issued now has eight visits to arrive: waiting
            // for it here was ~0.7 us per poll, 19 polls per block on the Gowalla shape)
            mine = polled != 0;
            polled = __hip_atomic_load(blk_flag + ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mine) *s_stop = 1;
        return true;
    };
    auto stale_read = [&](bool looked) -> bool {              // behind the barrier
        if (!looked) return false;
        const bool stop = *s_stop != 0;
        if (stop && tid == 0) __hip_atomic_store(blk_flag + ub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return stop;
    };

    int t = visit(i0);
    if (i0 < i1) { load_tile(t); store_tile(std::integral_constant<int, 0>()); }
    uint32_t tm_cur = tm_next;
    __syncthreads();
    int vi = i0;
    while (vi < i1) {
        {
            const int tn = vi + 1 < i1 ? visit_after(t) : t;
            load_tile(tn);
            __builtin_amdgcn_sched_barrier(0);
            one_visit(std::integral_constant<int, 0>(), t, tm_cur);
            const bool looked = stale_mark(vi - i0 + 1);
            store_tile(std::integral_constant<int, 1>());
            __syncthreads();
            tm_cur = tm_next; t = tn; ++vi;
            if (stale_read(looked)) break;
        }
        if (vi >= i1) break;
        {
            const int tn = vi + 1 < i1 ? visit_after(t) : t;
            load_tile(tn);
            __builtin_amdgcn_sched_barrier(0);
            one_visit(std::integral_constant<int, 1>(), t, tm_cur);
            const bool looked = stale_mark(vi - i0 + 1);
            store_tile(std::integral_constant<int, 0>());
            __syncthreads();
            tm_cur = tm_next; t = tn; ++vi;
            if (stale_read(looked)) break;
        }
    }
    for (int k = tid; k < kUsersPerBlock; k += THREADS) {
        const int qq = ub * kUsersPerBlock + k;
        const int ql2 = (REPAIR && rl.compact) ? ubv * kUsersPerBlock + k : qq;
        if (qq < U) {
            counts[(size_t)split * rl.stride + ql2] = (int32_t)min(s_cnt[k], (uint32_t)cap);
            if (s_cnt[k] > (uint32_t)cap) overflow[ovf_per_user ? qq : 0] = 1;        // full: her list was cut
        }
    }
    }   // segments
}

// The sampling pass on the same operand copies (k_score_sample_b's successor): class maxima of acc'' * f_u over a strided
// 1/2^s sample of the catalogue.  With the epilogue in the operands a score costs its share of the MFMAs and one v_max;
// masked items are poisoned at accumulator init.  The rows of a virtual tile are 2^s items apart: every unit is gathered
// by its own address (row and unit index of a thread's units are fixed; the window's base is wave-uniform).
template <int D, int KIND, bool REPAIR = false, bool HALF = false>
__global__ __launch_bounds__(512, D <= 64 ? 4 : 2) void k_score_sample_c(
    int U, int n_local, const uint4 *__restrict__ users_c, const uint4 *__restrict__ items_c,
    const float *__restrict__ sig_u, const uint32_t *__restrict__ mask_bits, const uint32_t *__restrict__ zero_word,
    int ublocks, float *__restrict__ maxima, int sample_log2, int merge_pairs,
    const int32_t *__restrict__ ub_map, const int32_t *__restrict__ n_ub_dev,
    const float *__restrict__ qpart = nullptr, int n_part = 0, uint32_t *__restrict__ qmax_bits = nullptr) {
    // (after k_eval_prologue_prep: the last block first turns its per-block item norms into max |q| -- k_tau reads it next)
    if (qpart && blockIdx.x == gridDim.x - 1) qmax_from_parts(qpart, n_part, qmax_bits);
    using C = std::conditional_t<HALF, StreamCfgH1<D>, StreamCfgC<D>>;      // HALF: the fp16 filter's copies (one MFMA per 16 k)
    constexpr int UR = HALF ? D / 8 : 2 * D / 8;              // 16-byte units per query row
    constexpr int THREADS = 512, NS = C::NS, RU = C::RU, TU = C::TU, LDU = C::LDU, REM = C::REM;
    const int kStep = 1 << sample_log2;
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *s_t = reinterpret_cast<uint4 *>(smem);                                      // [2][TU]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int uslot = wid * 32 + col;
    const int tiles_total = (n_local + kTileItems - 1) / kTileItems;
    const int n_pad = tiles_total * kTileItems;
    const int T = (tiles_total + kStep - 1) / kStep;          // windows = virtual tiles per user block
    int n_ub = ublocks;                                       // (repair round: see k_score_stream)
    long long G = gridDim.x;
    const long long b = blockIdx.x;
    if (REPAIR) {
        n_ub = *n_ub_dev;
        if (n_ub == 0) return;
        G = (long long)n_ub * G / ublocks;                     // (the maxima keep their layout: ranges as long as in the full launch)
        if (G < 1) G = 1;
        if (b >= G) return;
    }
    int S = (int)(0.6180339f * (float)T);                     // the visit order of k_score_stream
    S = S < 1 ? 1 : S;
    for (;; ++S) {
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T) * kStep; };
    auto visit_after = [&](int tile) { const int n = (tile >> sample_log2) + S; return (n >= T ? n - T : n) << sample_log2; };   // visit(i + 1) from visit(i)
    // this thread's units of a tile: (row, unit in row); the last round wraps (everybody loads, the owners store)
    int urow_[LDU], ucol_[LDU];
#pragma unroll
    for (int k = 0; k < LDU; ++k) {
        const int e = (tid + THREADS * k) % TU;
        urow_[k] = e / RU; ucol_[k] = e % RU;
    }
    const size_t mask_stride = mask_bits ? (size_t)U : 0;
    const long long W = (long long)n_ub * T;
    const long long w_end = W * (b + 1) / G;
    for (long long w = W * b / G; w < w_end;) {
    const int ubv = (int)(w / T), i0 = (int)(w - (long long)ubv * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ubv * T * G / W;
    while (W * (first + 1) / G <= (long long)ubv * T) ++first;
    while (W * first / G > (long long)ubv * T) --first;
    const int split = (int)(b - first);
    const int ub = REPAIR ? ub_map[ubv] : ubv;
    const int q = ub * kUsersPerBlock + uslot;
    const bool q_ok = q < U;
    __syncthreads();                                          // (the previous segment's readers are done with s_t)
    bf16x8 bhi[NS], blo[NS];
    {
        const uint4 *urow = users_c + (size_t)(q_ok ? q : 0) * UR;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = HALF ? v : urow[D / 8 + 2 * sI + h];
            if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[sI] = *reinterpret_cast<bf16x8 *>(&v);        // (HALF: eight fp16 numbers travelling in the bf16 type)
            blo[sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
    }
    const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;
    const bool tiny = score_uses_sig_u(KIND) && sig_u_tiny(su);
    const float fu = HALF ? query_factor_h<KIND>(su) : (KIND == MACR_SCORE_DIRECT_MINUS_BOTH && tiny) ? 1.0f : su;
    const bool ones_on = KIND != MACR_SCORE_NORMAL && h == 0 && !(KIND == MACR_SCORE_DIRECT_MINUS_BOTH && tiny);
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = ones_on ? 0x3f803f80u : 0u; ones.u[1] = ones_on ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    const bf16x8 bext = HALF ? bias_query_unit_h<KIND>(su, KIND != MACR_SCORE_NORMAL && h == 0) : ones.v;
    const uint32_t *my_mask = mask_bits ? mask_bits + (q_ok ? q : 0) : zero_word;

    uint4 stg[LDU];
    uint32_t tm_next = 0u;
    // item `row` of the virtual tile of window `tile >> s` = every kStep-th item of the window, phase = window index mod kStep
    auto load_tile = [&](int tile) {
        const int tu = __builtin_amdgcn_readfirstlane(tile);
        const int base = tu * kTileItems + ((tu >> sample_log2) & (kStep - 1));
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            const int it = min(base + kStep * urow_[k], n_pad - 1);            // (rows past the end: zero rows, masked below)
            stg[k] = items_c[(size_t)it * RU + ucol_[k]];
        }
        tm_next = my_mask[((size_t)tiles_total + (tu >> sample_log2)) * mask_stride];
    };
    auto store_tile = [&](auto BUF) {
        constexpr int buf = decltype(BUF)::value;
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            if (k + 1 < LDU || tid < REM) s_t[buf * TU + tid + THREADS * k] = stg[k];
        }
    };
    float cmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cmax[r] = -INFINITY;
    const float kNone = __builtin_nanf("");
    auto one_visit = [&](auto BUF, int t, uint32_t tm_cur) {
        constexpr int buf = decltype(BUF)::value;
        uint32_t tmask = tm_cur;
        const int valid = (n_local - t * kTileItems - ((t >> sample_log2) & (kStep - 1)) + kStep - 1) >> sample_log2;   // < 32 only in the last window
        if (valid < kTileItems) tmask |= valid > 0 ? ~0u << valid : ~0u;
        const uint32_t tmh = tmask >> (4 * h);
        const __bf16 *ua = reinterpret_cast<const __bf16 *>(s_t + buf * TU) + (size_t)col * (8 * RU) + 8 * h;
        f32x16 acc;
        // (poisoning costs 16 selects and an accumulator that starts in registers: only where some lane of the wave masks something
        // in this tile -- about every second visit on the Gowalla shape; 52 -> ~47 us for the pass)
        if (__builtin_amdgcn_ballot_w64(tmh != 0u)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = ((tmh >> ((r & 3) + 8 * (r >> 2))) & 1u) ? kNone : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        constexpr int CH = NS < 4 ? NS : 4;
        bf16x8 ae;
        if (KIND != MACR_SCORE_NORMAL)
            ae = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(s_t + buf * TU) + (size_t)col * (8 * RU) + (HALF ? D : 2 * D));
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += CH) {
            bf16x8 ah[CH], al[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                ah[j] = *reinterpret_cast<const bf16x8 *>(ua + 16 * (s0 + j));
                if (!HALF) al[j] = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * (s0 + j));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (HALF) {
                    acc = mfma_f16(ah[j], bhi[s0 + j], acc);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[j], bhi[s0 + j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[j], blo[s0 + j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[j], bhi[s0 + j], acc, 0, 0, 0);
                }
            }
            if (KIND != MACR_SCORE_NORMAL && s0 + CH >= NS) acc = HALF ? mfma_f16(ae, bext, acc) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) cmax[r] = fmaxf(cmax[r], acc[r]);        // (fmaxf drops the NaN of a masked item)
    };
    int vi = i0, t = visit(i0);
    if (vi < i1) { load_tile(t); store_tile(std::integral_constant<int, 0>()); }
    uint32_t tm_cur = tm_next;
    __syncthreads();
    while (vi < i1) {
        {
            const int tn = vi + 1 < i1 ? visit_after(t) : t;
            load_tile(tn);
            __builtin_amdgcn_sched_barrier(0);
            one_visit(std::integral_constant<int, 0>(), t, tm_cur);
            store_tile(std::integral_constant<int, 1>());
            __syncthreads();
            tm_cur = tm_next; t = tn; ++vi;
        }
        if (vi >= i1) break;
        {
            const int tn = vi + 1 < i1 ? visit_after(t) : t;
            load_tile(tn);
            __builtin_amdgcn_sched_barrier(0);
            one_visit(std::integral_constant<int, 1>(), t, tm_cur);
            store_tile(std::integral_constant<int, 0>());
            __syncthreads();
            tm_cur = tm_next; t = tn; ++vi;
        }
    }
    // Classes are merged in pairs (slots r and r + 8: disjoint item sets stay disjoint): 16 maxima per (split, query) for
    // k_tau to rank instead of 32 -- half its registers and scalar work; two of a query's K best classes share a pair ~once
    // in K (K - 1) / 2 / 160 = 1.2 cases at K = 20, which lowers the threshold by one rank of ~160.  Scores = acc'' * f_u.
    // (merge_pairs: the launcher's choice -- only where 16 per split leave several times K classes)
    if (q_ok && !merge_pairs) {
        float *o = maxima + ((size_t)split * U + q) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g)       // slots (r&3)+8(r>>2)+4h: four runs of four consecutive floats
            *reinterpret_cast<float4 *>(o + 8 * g + 4 * h) = make_float4(cmax[4 * g] * fu, cmax[4 * g + 1] * fu, cmax[4 * g + 2] * fu, cmax[4 * g + 3] * fu);
    }
    if (q_ok && merge_pairs) {
        float *o = maxima + ((size_t)split * U + q) * 16 + 8 * h;
        *reinterpret_cast<float4 *>(o) = make_float4(fmaxf(cmax[0], cmax[8]) * fu, fmaxf(cmax[1], cmax[9]) * fu, fmaxf(cmax[2], cmax[10]) * fu,
                                                     fmaxf(cmax[3], cmax[11]) * fu);
        *reinterpret_cast<float4 *>(o + 4) = make_float4(fmaxf(cmax[4], cmax[12]) * fu, fmaxf(cmax[5], cmax[13]) * fu, fmaxf(cmax[6], cmax[14]) * fu,
                                                         fmaxf(cmax[7], cmax[15]) * fu);
    }
    }   // segments
}

// ----------------------------------------------------------------------------
// k_score_stream_h: the listing pass of the fp16 filter.  Same work decomposition, lists, counters, stale-seed protocol
// and repair layout as k_score_stream_c (a block = one 256-query block x a scattered range of item tiles), but with one
// MFMA per 16 k the matrix cores are no longer what a visit costs -- the LDS reads of the item fragments and the barrier
// are -- so a wave keeps TWO 32-query groups in registers (64 queries; B operands 2 x D/2 bytes per lane) and every item
// fragment it reads serves two MFMAs: four waves cover the 256 queries of a block, each tile's fragments are read by four
// waves instead of eight -- LDS reads per 32 x 32 products: a quarter of k_score_stream_c's in count, an eighth in bytes.
// A block is those four waves and one tile per step (kHTiles = 1, four blocks per CU); with kHTiles = 2 it is eight waves
// whose halves take two DIFFERENT tiles of one step (two tiles staged per barrier; the two waves that share a query append
// to her list through the same LDS counter) -- the first form built, 20 % slower: see kHTiles.
// Measured on the Gowalla shape (15 424 queries x 40 981 items, d = 64; profiles/r06_eval_f16_filter.txt): 134 us under
// sampled thresholds, 123 under seeded ones (k_score_stream_c: 247 / 228).  With kHTiles = 2 (165-172 / 145-148), by ablation (-DMACR_ABL_H_*): without the
// epilogue 97 us, without epilogue and MFMAs 82 us, without the barrier no change -- the skeleton of a step (two tiles from
// global memory through registers into LDS, five fragment reads per wave, the visit arithmetic) is most of the pass, the 10
// MFMAs of a wave's step hide under it, and the epilogue is the rest.  The matrix cores alone would need 46 us.  The skeleton is
// a latency chain per step (load -> wait -> ds_write -> barrier -> fragment reads -> wait), two blocks per CU to overlap it:
// with epilogue, MFMAs and mask loads compiled out it takes 76-81 us, 77 with every tile load forced to hit the cache
// (-DMACR_ABL_H_SAMETILE: not a memory bound), 70-74 without the barrier as well.
// ----------------------------------------------------------------------------
template <int D, int KIND, bool REPAIR = false>
__global__ __launch_bounds__(StreamCfgH<D>::THREADS, D <= 64 ? 4 : 2) void k_score_stream_h(
    int U, int n_local, const uint4 *__restrict__ users_h, const uint4 *__restrict__ items_h,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits,
    const float *__restrict__ sig_u, float c_val, const float *__restrict__ c_dev,
    const uint32_t *__restrict__ mask_bits, const uint32_t *__restrict__ zero_word, int item_offset, int ublocks,
    const float *__restrict__ tau, uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow,
    int ovf_per_user, int32_t *blk_flag, const int32_t *__restrict__ ub_map, const int32_t *__restrict__ n_ub_dev, int slots_full) {
    using C = StreamCfgH<D>;
    constexpr int THREADS = C::THREADS, NS = C::NS, RU = C::RU, TU = C::TU, SU = C::SU, LDU = C::LDU;
    constexpr int kCheckTiles = 8;
    const float c = c_dev ? *c_dev : c_val;
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *s_t = reinterpret_cast<uint4 *>(smem);                                      // [2 buffers][2 tiles][TU]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)2 * SU * 16);       // [256]
    int *s_stop = reinterpret_cast<int *>(s_cnt + kUsersPerBlock);                     // [1]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int tslot = kHTiles == 2 ? __builtin_amdgcn_readfirstlane(wid >> 2) : 0, uw = wid & 3;      // this wave's tile of a step; its 64 queries
    const int T = (n_local + kTileItems - 1) / kTileItems;
    int n_ub = ublocks;
    long long G = gridDim.x;
    const long long b = blockIdx.x;
    RepairLayout rl = {false, 0, U};
    if (REPAIR) {
        n_ub = *n_ub_dev;
        if (n_ub == 0) return;
        rl = repair_layout(slots_full, U, n_ub);
        G = repair_grid(rl, G, ublocks, n_ub, T);
        if (b >= G) return;
    }
    int S = (int)(0.6180339f * (float)T);                     // the visit order of k_score_stream (scattered tile ranges)
    S = S < 1 ? 1 : S;
    for (;; ++S) {
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T); };
    auto visit_after = [&](int tile) { const int n = tile + S; return n >= T ? n - T : n; };
    const float qmax = __uint_as_float(*qmax_bits);
    const long long W = (long long)n_ub * T;
    const long long w_end = W * (b + 1) / G;
    const size_t mask_stride = mask_bits ? (size_t)U : 0;
    // (An XCD-aware decomposition -- block b lists only visits of XCD (b % 8)'s eighth of the visit sequence, so that the blocks
    // of an XCD stream the same 0.7 MB of tiles through its L2 -- was built and measured: 166 against 164 us, no gain; the pass
    // is not bound by where its tiles come from.  profiles/r06_eval_f16_filter.txt)
    for (long long w = W * b / G; w < w_end;) {
    const int ubv = (int)(w / T), i0 = (int)(w - (long long)ubv * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ubv * T * G / W;
    while (W * (first + 1) / G <= (long long)ubv * T) ++first;
    while (W * first / G > (long long)ubv * T) --first;
    const int split = (int)(b - first);
    const int ub = REPAIR ? ub_map[ubv] : ubv;
    __syncthreads();                                          // (the previous segment's readers are done with s_t and s_cnt)
    for (int k = tid; k < kUsersPerBlock; k += THREADS) s_cnt[k] = 0u;
    if (tid == 0) *s_stop = 0;
    bf16x8 bq[2][NS], bext[2];
    float thr[2], fu[2];
    uint32_t my_list[2], my_q[2];                             // list index (units of cap) / clamped query index: addresses are formed where used
    uint32_t *my_cnt[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int uslot = uw * 64 + g * 32 + col, q = ub * kUsersPerBlock + uslot;
        const bool q_ok = q < U;
        const uint4 *urow = users_h + (size_t)(q_ok ? q : 0) * (D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h];
            if (!q_ok) v = make_uint4(0u, 0u, 0u, 0u);
            bq[g][sI] = *reinterpret_cast<bf16x8 *>(&v);
        }
        const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;
        const bool tiny = KIND == MACR_SCORE_RUBI_BOTH && sig_u_tiny(su);
        fu[g] = query_factor_h<KIND>(su);                     // listed score = acc'' * fu
        bext[g] = bias_query_unit_h<KIND>(su, KIND != MACR_SCORE_NORMAL && h == 0);
        float tau_s = __builtin_nanf("");
        if (q_ok) tau_s = tau[q] - 1.01f * filter_margin_h(D, unorm[q], qmax, c) * fu[g];      // (the margin of acc'' times the score's outer factor)
        thr[g] = tau_s / fu[g];
        if (tiny) thr[g] = q_ok ? -INFINITY : thr[g];        // (scores of the order of sig_u: everything is a candidate)
        my_list[g] = (uint32_t)((size_t)split * rl.stride + ((REPAIR && rl.compact) ? ubv * kUsersPerBlock + uslot : (q_ok ? q : 0)));
        my_q[g] = mask_bits ? (uint32_t)(q_ok ? q : 0) : 0u;
        my_cnt[g] = &s_cnt[uslot];
    }
    const int id_lane = item_offset + 4 * h;                  // id of accumulator slot r: tile * 32 + (r & 3) + 8 * (r >> 2) + id_lane

    uint4 stg[LDU];
    uint32_t tm_next[2] = {0u, 0u};
    // Tiles (ta, tb) of the next step into registers (unit e of [tile A | tile B] by thread e mod 512; the last round wraps:
    // everybody loads, the owners store), then the mask words of THIS wave's tile.  (LDS-DMA -- global_load_lds_dwordx4
    // with scalar bases, no registers -- was built and measured 20 us slower for the pass: profiles/r06_eval_f16_filter.txt.)
    auto load_step = [&](int ta, int tb) {
        const int tau_ = __builtin_amdgcn_readfirstlane(ta), tbu = __builtin_amdgcn_readfirstlane(tb);
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            int e = tid + THREADS * k;
            if (k + 1 == LDU && e >= SU) e -= SU;              // (the wrapped loads of the last round: any valid unit)
            const bool second = kHTiles == 2 && e >= TU;
            stg[k] = items_h[(size_t)(second ? tbu : tau_) * TU + (second ? e - TU : e)];
        }
        const int mine = tslot ? tbu : tau_;
#ifndef MACR_ABL_H_NOMASK
        const uint32_t *mrow = (mask_bits ? mask_bits : zero_word) + (size_t)mine * mask_stride;
        tm_next[0] = mrow[my_q[0]];
        tm_next[1] = mrow[my_q[1]];
#else
        tm_next[0] = zero_word[mine & 0]; tm_next[1] = tm_next[0];
#endif
    };
    auto store_step = [&](auto BUF) {
        constexpr int buf = decltype(BUF)::value;
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            if (k + 1 < LDU || tid < SU - THREADS * (LDU - 1)) s_t[buf * SU + tid + THREADS * k] = stg[k];
        }
    };
    // one step: this wave's tile t (t < 0: none -- the odd tile out of a range) sits in half `tslot` of buffer BUF
    auto one_step = [&](auto BUF, int t, uint32_t tm0, uint32_t tm1) {
        constexpr int buf = decltype(BUF)::value;
        const __bf16 *ua = reinterpret_cast<const __bf16 *>(s_t + buf * SU + tslot * TU) + (size_t)col * (8 * RU) + 8 * h;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // (fragments two k-slabs at a time at d <= 64: with four, the kernel needs 132 registers and the spill's reload waits for
        // the tile prefetch -- vmcnt counts scratch and global loads alike)
        constexpr int CH = NS < 2 ? NS : (D <= 64 ? 2 : 4);
        bf16x8 ae;
        if (KIND != MACR_SCORE_NORMAL) ae = *reinterpret_cast<const bf16x8 *>(ua - 8 * h + D);      // (the bias unit: lanes of both halves read k < 8)
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += CH) {
            bf16x8 af[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) af[j] = *reinterpret_cast<const bf16x8 *>(ua + 16 * (s0 + j));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; ++j) {                    // (two independent accumulators: back to back without the dependent-issue gap)
#ifndef MACR_ABL_H_NOMFMA
                acc0 = mfma_f16(af[j], bq[0][s0 + j], acc0);
                acc1 = mfma_f16(af[j], bq[1][s0 + j], acc1);
#else
                acc0[j] += (float)af[j][0] * (float)bq[0][s0 + j][0];
                acc1[j] += (float)af[j][1] * (float)bq[1][s0 + j][1];
#endif
            }
            if (KIND != MACR_SCORE_NORMAL && s0 + CH >= NS) {
                acc0 = mfma_f16(ae, bext[0], acc0);
                acc1 = mfma_f16(ae, bext[1], acc1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int valid = t < 0 ? 0 : n_local - t * kTileItems;           // < 32 only in the last tile of the shard
        const uint32_t tailm = valid < kTileItems ? ~0u << (valid > 0 ? valid : 0) : 0u;
        const int id0 = t * kTileItems + id_lane;
        // Epilogue: per score one v_cmp + one s_cbranch (k_score_stream_c's), a hit costs eight vector instructions (mask bit,
        // LDS counter, key, store).  With one MFMA per 16 k those compare/branch pairs -- ~18 cycles each: the scalar branch
        // waits for the vector compare -- were 85 of the pass's 181 us (ablation: profiles/r06_eval_f16_filter.txt), so four scores
        // (one row quad of the tile) are first tested together: v_max3 + v_max + one pair, and only a quad somebody passes is
        // looked at score by score -- 2/3 of the quads under sampled thresholds (~176 listed per query), 1/6 under seeded ones.
        // (A branch-free form -- per-lane hit words by v_cmp + v_addc, one counter bump per lane and tile, a scalar loop
        // over the slots hit with a wave-uniform register index -- was built and measured: 195 / 178 us sampled / seeded
        // against 178 / 176 for the plain pairs; the loop's ~35 cycles per slot hit cost what the branches had.)
        auto epilogue = [&](const f32x16 &acc, uint32_t tm_cur, float thr_g, float fu_g, uint32_t *cnt_g, uint32_t list_g) {
            const uint32_t tmh = (tm_cur | tailm) >> (4 * h);  // bit (r & 3) + 8 (r >> 2): this lane's row of slot r is masked / past the end
#pragma unroll
            for (int r4 = 0; r4 < 16; r4 += 4) {
                const float quad = fmaxf(fmaxf(fmaxf(acc[r4], acc[r4 + 1]), acc[r4 + 2]), acc[r4 + 3]);       // (fmaxf drops a NaN: as the compare below would)
                if (!__builtin_amdgcn_ballot_w64(quad >= thr_g)) continue;
#pragma unroll
                for (int r = r4; r < r4 + 4; ++r) {
                    const bool above = acc[r] >= thr_g;
                    if (__builtin_amdgcn_ballot_w64(above)) {  // wave-uniform: some lane's score r passes
                        const int rbit = (r & 3) + 8 * (r >> 2);
                        uint32_t m = tmh;
                        asm volatile("" : "+v"(m));            // (the mask test belongs in here, not in front of the branch)
                        if (above && !((m >> rbit) & 1u)) {
                            const uint32_t pos = atomicAdd(cnt_g, 1u);     // (keeps counting past cap: the segment's end flags it)
                            if (pos < (uint32_t)cap) lists[(size_t)list_g * cap + pos] = make_key(acc[r] * fu_g, id0 + rbit);
                        }
                    }
                }
            }
        };
#ifndef MACR_ABL_H_NOEPI
        epilogue(acc0, tm0, thr[0], fu[0], my_cnt[0], my_list[0]);
        epilogue(acc1, tm1, thr[1], fu[1], my_cnt[1], my_list[1]);
#else
        {
            float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, acc0[r]); m1 = fmaxf(m1, acc1[r]); }
            if (m0 == 1.2345e30f || m1 == 1.2345e30f) epilogue(acc0, tm0, thr[0], fu[0], my_cnt[0], my_list[0]);
        }
#endif
    };
    // stale seeds (k_score_stream_c): `done` = tiles of this segment visited by the block so far
    int polled = 0;
    auto stale_mark = [&](int done_before, int done) -> bool {
        if (!blk_flag) return false;
        auto crossed = [&](int v) { return done_before < v && done >= v; };
        const bool check = crossed(2) || crossed(kCheckTiles) || crossed(4 * kCheckTiles);
        if (!(check || (done >> 3) != (done_before >> 3))) return false;
        bool mine = false;
        if (check) {
            uint32_t a = tslot == 0 ? s_cnt[uw * 64 + lane] : 0u;                      // appended so far to the lists of this wave's 64 queries (by both waves that share them)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, kWave);
            const float usable = 64.f * (float)(done * kTileItems) * (float)(kSelRegs * 64) / (float)n_local;
            mine = tslot == 0 && (float)a > usable + 128.f;
        } else if (tid == 0) {
            mine = polled != 0;
            polled = __hip_atomic_load(blk_flag + ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mine) *s_stop = 1;
        return true;
    };
    auto stale_read = [&](bool looked) -> bool {              // behind the barrier
        if (!looked) return false;
        const bool stop = *s_stop != 0;
        if (stop && tid == 0) __hip_atomic_store(blk_flag + ub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return stop;
    };

    // steps of two tiles: (ta, tb) = (visit(vi), visit(vi + 1)); an odd range ends with a step whose second tile is none
    int vi = i0;
    auto second_of = [&](int first_tile, int v) { return (kHTiles == 2 && v + 1 < i1) ? visit_after(first_tile) : -1; };
    auto first_after = [&](int a_, int b_, int v) { return v + kHTiles < i1 ? visit_after(b_ < 0 ? a_ : b_) : a_; };
    int ta = visit(i0), tb = second_of(ta, vi);
    if (vi < i1) { load_step(ta, tb < 0 ? ta : tb); store_step(std::integral_constant<int, 0>()); }
    uint32_t tm0 = tm_next[0], tm1 = tm_next[1];
    __syncthreads();
    while (vi < i1) {
        {
            const int na = first_after(ta, tb, vi), nb = second_of(na, vi + kHTiles);
            load_step(na, nb < 0 ? na : nb);
            __builtin_amdgcn_sched_barrier(0);
            one_step(std::integral_constant<int, 0>(), tslot ? tb : ta, tm0, tm1);
            const int done = min(vi + kHTiles, i1) - i0;
            const bool looked = stale_mark(vi - i0, done);
            store_step(std::integral_constant<int, 1>());
#ifndef MACR_ABL_H_NOBARRIER
            __syncthreads();
#endif
            tm0 = tm_next[0]; tm1 = tm_next[1]; ta = na; tb = nb; vi += kHTiles;
            if (stale_read(looked)) break;
        }
        if (vi >= i1) break;
        {
            const int na = first_after(ta, tb, vi), nb = second_of(na, vi + kHTiles);
            load_step(na, nb < 0 ? na : nb);
            __builtin_amdgcn_sched_barrier(0);
            one_step(std::integral_constant<int, 1>(), tslot ? tb : ta, tm0, tm1);
            const int done = min(vi + kHTiles, i1) - i0;
            const bool looked = stale_mark(vi - i0, done);
            store_step(std::integral_constant<int, 0>());
#ifndef MACR_ABL_H_NOBARRIER
            __syncthreads();
#endif
            tm0 = tm_next[0]; tm1 = tm_next[1]; ta = na; tb = nb; vi += kHTiles;
            if (stale_read(looked)) break;
        }
    }
    for (int k = tid; k < kUsersPerBlock; k += THREADS) {
        const int qq = ub * kUsersPerBlock + k;
        const int ql2 = (REPAIR && rl.compact) ? ubv * kUsersPerBlock + k : qq;
        if (qq < U) {
            counts[(size_t)split * rl.stride + ql2] = (int32_t)min(s_cnt[k], (uint32_t)cap);
            if (s_cnt[k] > (uint32_t)cap) overflow[ovf_per_user ? qq : 0] = 1;        // full: her list was cut
        }
    }
    }   // segments
}

// Test-only (macr_test_bf16_scores): the SCORE k_score_stream_c lists for every (query, item) pair -- its MFMA sequence on
// the copies k_bf16_prep_c wrote, times the query's factor -- and the margin the filter grants each query.
// HALF: the fp16 filter's copies and k_score_stream_h's sequence (one MFMA per 16 k, then the bias slab).
template <int D, int KIND, bool HALF = false>
__global__ __launch_bounds__(64) void k_test_bf16_scores(int U, int N, const uint4 *__restrict__ users_c, const uint4 *__restrict__ items_c,
                                                         const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits,
                                                         const float *__restrict__ sig_u, float c, float *__restrict__ out,
                                                         float *__restrict__ margin) {
    constexpr int RU = HALF ? D / 8 + 1 : StreamCfgC<D>::RU;
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    const int item = min((int)blockIdx.x * 32 + col, N - 1), user = min((int)blockIdx.y * 32 + col, U - 1);
    const uint4 *irow = items_c + (size_t)item * RU, *urow = users_c + (size_t)user * (HALF ? D / 8 : 2 * D / 8);
    const float su = score_uses_sig_u(KIND) ? sig_u[user] : 1.0f;
    const bool tiny = score_uses_sig_u(KIND) && sig_u_tiny(su);
    const float fu = HALF ? query_factor_h<KIND>(su) : (KIND == MACR_SCORE_DIRECT_MINUS_BOTH && tiny) ? 1.0f : su;
    const bool ones_on = KIND != MACR_SCORE_NORMAL && h == 0 && !(KIND == MACR_SCORE_DIRECT_MINUS_BOTH && tiny);
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = ones_on ? 0x3f803f80u : 0u; ones.u[1] = ones_on ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    if (HALF) ones.v = bias_query_unit_h<KIND>(su, KIND != MACR_SCORE_NORMAL && h == 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int sI = 0; sI < D / 16; ++sI) {
        uint4 v;
        v = irow[2 * sI + h];          const bf16x8 ah = *reinterpret_cast<bf16x8 *>(&v);
        v = urow[2 * sI + h];          const bf16x8 bh = *reinterpret_cast<bf16x8 *>(&v);
        if (HALF) {
            acc = mfma_f16(ah, bh, acc);
        } else {
            v = irow[D / 8 + 2 * sI + h];  const bf16x8 al = *reinterpret_cast<bf16x8 *>(&v);
            v = urow[D / 8 + 2 * sI + h];  const bf16x8 bl = *reinterpret_cast<bf16x8 *>(&v);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    if (KIND != MACR_SCORE_NORMAL) {
        uint4 v = irow[HALF ? D / 8 : 2 * D / 8];
        acc = HALF ? mfma_f16(*reinterpret_cast<bf16x8 *>(&v), ones.v, acc)
                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&v), ones.v, acc, 0, 0, 0);
    }
    const int u = (int)blockIdx.y * 32 + col;
    if (u >= U) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int it = (int)blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (it < N) out[(size_t)u * N + it] = acc[r] * fu;
    }
    if (blockIdx.x == 0 && h == 0) margin[u] = filter_margin_of(HALF ? 1 : 0, D, unorm[u], __uint_as_float(*qmax_bits), c) * (HALF ? fu : 1.0f);
}

// The listing pass of k_score_stream_b for up to kMaxSweep values of c at once (macr_score_topk_sweep under the bf16
// filter): one staging of the item tiles, one set of MFMAs and one barrier per tile serve every value; per (score, c)
// there remain the test on the raw product and, for the hits, the epilogue and the append into that c's lists.
template <int D, int KIND>
__global__ __launch_bounds__(512, D <= 64 ? 4 : 2) void k_score_stream_bs(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items_bf,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits,
    const float *__restrict__ sig_u, const float *__restrict__ sig_i, const float *__restrict__ c_dev,
    const uint32_t *__restrict__ mask_bits, int item_offset, int ublocks, int cap, SweepArgs sw) {
    using C = StreamCfgB<D>;
    constexpr int NC = kMaxSweep, LDU = (C::UNITS + 511) / 512;
    constexpr int RSB = C::RSB, NS = C::NS;
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *s_a = reinterpret_cast<__bf16 *>(smem);                                   // [2][32][RSB]
    float *s_sig = reinterpret_cast<float *>(smem + (size_t)2 * kTileItems * RSB * 2);  // [2][32]
    float *s_y = s_sig + 2 * kTileItems;                                               // [2][32]  filter_y(sig_i)
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_y + 2 * kTileItems);            // [NC][256]
    int *s_slow = reinterpret_cast<int *>(s_cnt + NC * kUsersPerBlock);               // [2]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int uslot = wid * 32 + col;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    const long long G = gridDim.x, b = blockIdx.x;
    int S = (int)(0.6180339f * (float)T);                     // the visit order of k_score_stream
    S = S < 1 ? 1 : S;
    for (;; ++S) {
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T); };
    auto visit_after = [&](int tile) { const int n = tile + S; return n >= T ? n - T : n; };
    const float qmax = __uint_as_float(*qmax_bits);
    const long long W = (long long)ublocks * T;
    const long long w_end = W * (b + 1) / G;
    for (long long w = W * b / G; w < w_end;) {
    const int ub = (int)(w / T), i0 = (int)(w - (long long)ub * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ub * T * G / W;
    while (W * (first + 1) / G <= (long long)ub * T) ++first;
    while (W * first / G > (long long)ub * T) --first;
    const int split = (int)(b - first);
    const int q = ub * kUsersPerBlock + uslot;
    const bool q_ok = q < U;

    for (int k = tid; k < NC * kUsersPerBlock; k += 512) s_cnt[k] = 0u;
    bf16x8 bhi[NS], blo[NS];
    {
        const uint4 *urow = users_bf + (size_t)(q_ok ? q : 0) * (2 * D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
            if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[sI] = *reinterpret_cast<bf16x8 *>(&v);
            blo[sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
    }
    const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;
    const bool wave_slow = KIND == MACR_SCORE_RUBI_BOTH && __any(q_ok && !(su > 1e-30f));
    const float un = q_ok ? unorm[q] : 0.f;
    // per value of c: threshold less the margin, and the test on the raw product (filter_y): acc >= fma(fx, Y_i, fz)
    float c_g[NC], tau_g[NC], fx[NC], fz[NC];
#pragma unroll
    for (int g = 0; g < NC; ++g) {
        c_g[g] = g < sw.n_c ? c_dev[g] : 0.f;
        tau_g[g] = (g < sw.n_c && q_ok) ? sw.tau[g][q] - 1.01f * filter_margin(D, un, qmax, c_g[g]) : __builtin_nanf("");
        fx[g] = 0.f; fz[g] = tau_g[g];
        if (KIND == MACR_SCORE_RUBI_BOTH) { fx[g] = tau_g[g] / su; fz[g] = c_g[g]; }
        else if (KIND == MACR_SCORE_RUBI) { fx[g] = tau_g[g]; fz[g] = c_g[g]; }
        else if (KIND == MACR_SCORE_DIRECT_MINUS) { fx[g] = c_g[g]; }
        else if (KIND == MACR_SCORE_DIRECT_MINUS_BOTH) { fx[g] = c_g[g] * su; }
    }

    int vi = i0, t = visit(i0);
    uint4 stg[LDU];
    float sg = 0.f;
    uint32_t tm_next = 0u;
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            const int e = tid + 512 * k, row = (e / (2 * D / 8)) & (kTileItems - 1), c8 = e % (2 * D / 8);
            const int it = min(tile * kTileItems + row, n_local - 1);
            stg[k] = items_bf[(size_t)it * (2 * D / 8) + c8];
        }
        sg = sig_i[min(tile * kTileItems + (tid & (kTileItems - 1)), n_local - 1)];
        tm_next = (mask_bits && q_ok) ? mask_bits[(size_t)tile * U + q] : 0u;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            const int e = tid + 512 * k, row = e / (2 * D / 8), c8 = e % (2 * D / 8);
            if (row < kTileItems)
                *reinterpret_cast<uint4 *>(s_a + ((size_t)buf * kTileItems + row) * RSB + 8 * c8) = stg[k];
        }
        asm volatile("" : "+v"(sg));
        if (tid < 64) {
            const bool tiny = !(sg > 1e-30f);
            if (tid < kTileItems) { s_sig[buf * kTileItems + tid] = sg; s_y[buf * kTileItems + tid] = filter_y<KIND>(sg); }
            const bool any_tiny = __any(tiny);
            if (tid == 0) s_slow[buf] = any_tiny ? 1 : 0;
        }
    };

    int buf = 0;
    if (vi < i1) { load_tile(t); store_tile(0); }
    uint32_t tm_cur = tm_next;
    __syncthreads();
    while (vi < i1) {
        const bool has_next = vi + 1 < i1;
        const int tn = has_next ? visit_after(t) : t;
        if (has_next) load_tile(tn);

        const int gid0 = t * kTileItems + item_offset;
        const int valid = n_local - t * kTileItems;
        const uint32_t tmask = tm_cur | (valid < kTileItems ? (valid > 0 ? ~0u << valid : ~0u) : 0u);

        const __bf16 *ua = s_a + ((size_t)buf * kTileItems + col) * RSB + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[sI], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[sI], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[sI], acc, 0, 0, 0);
        }
        const bool slow = wave_slow || s_slow[buf] != 0;      // wave-uniform, practically never
        float yi[16];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<float4 *>(yi + 4 * k) = *reinterpret_cast<const float4 *>(s_y + buf * kTileItems + 8 * k + 4 * h);
#pragma unroll
        for (int g = 0; g < NC; ++g) {
            if (g >= sw.n_c) break;                           // wave-uniform
            uint32_t hit = 0u;
            if (!slow) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hit |= __ballot(acc[r] >= fmaf(fx[g], yi[r], fz[g])) ? 1u << r : 0u;
            } else {
                hit = 0xffffu;
            }
            if (hit) {
                uint64_t *lst = sw.lists[g] + ((size_t)split * U + (q_ok ? q : 0)) * cap;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!((hit >> r) & 1u)) continue;         // wave-uniform
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    bool pass = slow || acc[r] >= fmaf(fx[g], yi[r], fz[g]);
                    pass = pass && ((tmask >> row) & 1u) == 0u;
                    if (pass) {
                        const float v = score_epilogue<KIND>(acc[r], c_g[g], s_sig[buf * kTileItems + row], su);
                        if (v >= tau_g[g]) {
                            const uint32_t pos = atomicAdd(&s_cnt[g * kUsersPerBlock + uslot], 1u);
                            if (pos < (uint32_t)cap) lst[pos] = make_key(v, gid0 + row);
                            else {                            // full: that c's fallback kernel ranks; stop listing for her
                                *sw.overflow[g] = 1;
                                tau_g[g] = INFINITY; fz[g] = INFINITY;
                                if (KIND == MACR_SCORE_RUBI_BOTH || KIND == MACR_SCORE_RUBI) fx[g] = INFINITY;
                            }
                        }
                    }
                }
            }
        }
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
        tm_cur = tm_next;
        buf ^= 1;
        t = tn; ++vi;
    }
    for (int k = tid; k < kUsersPerBlock; k += 512) {
        const int qq = ub * kUsersPerBlock + k;
#pragma unroll
        for (int g = 0; g < NC; ++g)
            if (g < sw.n_c && qq < U) sw.counts[g][(size_t)split * U + qq] = (int32_t)min(s_cnt[g * kUsersPerBlock + k], (uint32_t)cap);
    }
    }   // segments
}

// The sampling pass of k_score_stream (MODE = max) on the bf16 copies: per user and class the largest score_bf16 among
// the sampled unmasked items.  K classes with maxima >= t hold K distinct items whose fp32 scores are >= t - margin_u:
// k_tau takes the K-th largest maximum and subtracts the margin (its `unorm` argument), which makes tau a valid fp32
// threshold again; the listing pass then works as after an fp32 sampling pass.
template <int D, int KIND, bool REPAIR = false>
__global__ __launch_bounds__(512, D <= 64 ? 4 : 2) void k_score_sample_b(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items_bf,
    const float *__restrict__ sig_u, const float *__restrict__ sig_i, float c_val, const float *__restrict__ c_dev,
    const uint32_t *__restrict__ mask_bits, int ublocks, float *__restrict__ maxima, int sample_log2,
    const int32_t *__restrict__ ub_map, const int32_t *__restrict__ n_ub_dev) {
    using C = StreamCfgB<D>;
    constexpr int LDU = (C::UNITS + 511) / 512;
    const float c = c_dev ? *c_dev : c_val;
    constexpr int RSB = C::RSB, NS = C::NS;
    const int kStep = 1 << sample_log2;
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *s_a = reinterpret_cast<__bf16 *>(smem);                                   // [2][32][RSB]
    float *s_sig = reinterpret_cast<float *>(smem + (size_t)2 * kTileItems * RSB * 2);  // [2][32]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int uslot = wid * 32 + col;
    const int tiles_total = (n_local + kTileItems - 1) / kTileItems;
    const int T = (tiles_total + kStep - 1) / kStep;          // windows = virtual tiles per user block
    int n_ub = ublocks;                                       // (repair round: see k_score_stream)
    long long G = gridDim.x;
    const long long b = blockIdx.x;
    if (REPAIR) {
        n_ub = *n_ub_dev;
        if (n_ub == 0) return;
        G = (long long)n_ub * G / ublocks;                     // (the maxima keep their layout: ranges as long as in the full launch)
        if (G < 1) G = 1;
        if (b >= G) return;
    }
    int S = (int)(0.6180339f * (float)T);                     // the visit order of k_score_stream
    S = S < 1 ? 1 : S;
    for (;; ++S) {
        int x = S, y = T;
        while (y) { const int r = x % y; x = y; y = r; }
        if (x == 1) break;
    }
    auto visit = [&](int i) { return (int)(((unsigned long long)i * (unsigned)S) % (unsigned)T) * kStep; };
    auto visit_after = [&](int tile) { const int n = (tile >> sample_log2) + S; return (n >= T ? n - T : n) << sample_log2; };
    const long long W = (long long)n_ub * T;
    const long long w_end = W * (b + 1) / G;
    for (long long w = W * b / G; w < w_end;) {
    const int ubv = (int)(w / T), i0 = (int)(w - (long long)ubv * T);
    const int i1 = (int)min((long long)T, i0 + (w_end - w));
    w += i1 - i0;
    long long first = (long long)ubv * T * G / W;
    while (W * (first + 1) / G <= (long long)ubv * T) ++first;
    while (W * first / G > (long long)ubv * T) --first;
    const int split = (int)(b - first);
    const int ub = REPAIR ? ub_map[ubv] : ubv;
    const int q = ub * kUsersPerBlock + uslot;
    const bool q_ok = q < U;

    bf16x8 bhi[NS], blo[NS];
    {
        const uint4 *urow = users_bf + (size_t)(q_ok ? q : 0) * (2 * D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
            if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[sI] = *reinterpret_cast<bf16x8 *>(&v);
            blo[sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
    }
    const float su = (score_uses_sig_u(KIND) && q_ok) ? sig_u[q] : 1.0f;

    int vi = i0, t = visit(i0);
    uint4 stg[LDU];
    float sg = 0.f;
    uint32_t tm_next = 0u;
    // item j of the virtual tile of window `tile >> s` = every kStep-th item of the window, phase = window index mod kStep
    auto sampled_item = [&](int tile, int row) { return min(tile * kTileItems + kStep * row + ((tile >> sample_log2) & (kStep - 1)), n_local - 1); };
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            const int e = tid + 512 * k, row = (e / (2 * D / 8)) & (kTileItems - 1), c8 = e % (2 * D / 8);
            stg[k] = items_bf[(size_t)sampled_item(tile, row) * (2 * D / 8) + c8];
        }
        if (score_uses_sig_i(KIND)) sg = sig_i[sampled_item(tile, tid & (kTileItems - 1))];
        tm_next = (mask_bits && q_ok) ? mask_bits[((size_t)tiles_total + (tile >> sample_log2)) * U + q] : 0u;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int k = 0; k < LDU; ++k) {
            asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
            const int e = tid + 512 * k, row = e / (2 * D / 8), c8 = e % (2 * D / 8);
            if (row < kTileItems)
                *reinterpret_cast<uint4 *>(s_a + ((size_t)buf * kTileItems + row) * RSB + 8 * c8) = stg[k];
        }
        if (score_uses_sig_i(KIND)) {
            asm volatile("" : "+v"(sg));
            if (tid < kTileItems) s_sig[buf * kTileItems + tid] = sg;
        }
    };

    float cmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cmax[r] = -INFINITY;
    int buf = 0;
    if (vi < i1) { load_tile(t); store_tile(0); }
    uint32_t tm_cur = tm_next;
    __syncthreads();
    const float kNone = __builtin_nanf("");
    while (vi < i1) {
        const bool has_next = vi + 1 < i1;
        const int tn = has_next ? visit_after(t) : t;
        if (has_next) load_tile(tn);

        uint32_t tmask = tm_cur;
        const int valid = (n_local - t * kTileItems - ((t >> sample_log2) & (kStep - 1)) + kStep - 1) >> sample_log2;   // < 32 only in the last window
        if (valid < kTileItems) tmask |= valid > 0 ? ~0u << valid : ~0u;

        const __bf16 *ua = s_a + ((size_t)buf * kTileItems + col) * RSB + 8 * h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = ((tmask >> ((r & 3) + 8 * (r >> 2) + 4 * h)) & 1u) ? kNone : 0.f;
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[sI], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[sI], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[sI], acc, 0, 0, 0);
        }
        float sgi[16];
        if (score_uses_sig_i(KIND)) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(sgi + 4 * g) = *reinterpret_cast<const float4 *>(s_sig + buf * kTileItems + 8 * g + 4 * h);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if (score_uses_sig_i(KIND)) v = score_epilogue<KIND>(v, c, sgi[r], su);
            cmax[r] = fmaxf(cmax[r], v);                      // (fmaxf drops the NaN of a masked item)
        }
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
        tm_cur = tm_next;
        buf ^= 1;
        t = tn; ++vi;
    }
    if (q_ok) {
        float *o = maxima + ((size_t)split * U + q) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g)       // slots (r&3)+8(r>>2)+4h: four runs of four consecutive floats
            *reinterpret_cast<float4 *>(o + 8 * g + 4 * h) = make_float4(cmax[4 * g], cmax[4 * g + 1], cmax[4 * g + 2], cmax[4 * g + 3]);
    }
    }   // segments
}

constexpr int kSeedWidth = MACR_SEED_WIDTH;
static_assert(kSeedWidth == 32, "one 32-lane half per user");
template <int D, int KIND>
__device__ __forceinline__ void tau_seed_block(int blk, int U, int n_local, const float *__restrict__ users_tab,
                                               const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                               const float *__restrict__ sig_u, const float *__restrict__ sig_i,
                                               float c_val, const float *__restrict__ c_dev,
                                               const uint32_t *__restrict__ mask_bits, int item_offset, int K,
                                               const int32_t *__restrict__ seed, float *__restrict__ tau,
                                               const int32_t *__restrict__ mask_ptr = nullptr, const int32_t *__restrict__ mask_idx = nullptr) {
    const float c = c_dev ? *c_dev : c_val;
    const int q = blk * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (q >= U) return;                                   // whole 32-lane halves leave together
    const int it = seed[(size_t)q * kSeedWidth + l] - item_offset;
    bool ok = it >= 0 && it < n_local;
    if (ok && mask_ptr) {
        // a masked item is no candidate.  Looked up in the query's ascending train list (a few lines the 32 lanes of the half
        // share) rather than in the (tile, query) bitmap: one word per seed of a 79 MB array at the Gowalla shape -- 32 cold
        // lines per query after a log interval of training, where the list is one to four
        const int gid = it + item_offset;
        int lo = mask_ptr[q], hi = mask_ptr[q + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; const int v = mask_idx[mid]; if (v < gid) lo = mid + 1; else hi = mid; }
        ok = !(lo < mask_ptr[q + 1] && mask_idx[lo] == gid);
    } else if (ok && mask_bits) ok = ((mask_bits[(size_t)(it >> 5) * U + q] >> (it & 31)) & 1u) == 0u;
    float v = -INFINITY;
    if (ok) {
        const float *ur = users_tab + (size_t)(user_ids ? user_ids[q] : q) * D, *ir = items + (size_t)it * D;
        float acc = 0.f;
#pragma unroll 8
        for (int k4 = 0; k4 < D / 4; ++k4) {
            const float4 a = ld4(ur + 4 * k4), b = ld4(ir + 4 * k4);
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
        v = acc;
        if (score_uses_sig_i(KIND)) v = score_epilogue<KIND>(v, c, sig_i[it], score_uses_sig_u(KIND) ? sig_u[q] : 1.0f);
        ok = v == v;                                      // NaN: no bound from this seed
    }
    // rotation inside the half: drop repetitions of an earlier lane's id, then rank the good seeds by score
    const int half = threadIdx.x & 32;
#pragma unroll
    for (int m = 1; m < 32; ++m) {
        const int o = (l + m) & 31;
        const int oit = __shfl(it, half | o, kWave);
        const bool ook = __shfl((int)ok, half | o, kWave) != 0;     // the other's state BEFORE this loop drops anything:
        if (ook && oit == it && o < l) ok = false;                   // of equal ids the lowest lane survives (it never drops)
    }
    int rank = 0;
#pragma unroll
    for (int m = 1; m < 32; ++m) {
        const int o = (l + m) & 31;
        const float ov = __shfl(v, half | o, kWave);
        const bool ook = __shfl((int)ok, half | o, kWave) != 0;
        rank += (ook && (ov > v || (ov == v && o < l))) ? 1 : 0;
    }
    const uint64_t good = __ballot(ok) >> half & 0xffffffffull;
    const uint64_t kth = __ballot(ok && rank == K - 1) >> half & 0xffffffffull;       // exactly one lane if >= K good seeds
    float t = __shfl(v, half | (kth ? __builtin_ctzll(kth) : 0), kWave);
    // Two ulps of slack below the K-th seed score.  The argument above rests on this kernel's scalar fmaf chain being
    // bit-identical to the MFMA accumulation of the listing pass (it is on gfx950: tested); should another compiler or
    // chip round one of them differently by an ulp, a threshold set EXACTLY at a seed's score would list fewer than K
    // items for that user and nothing downstream would notice.  Any lower threshold is as valid, it only lengthens lists.
    const uint32_t ot = f32_orderable(t);
    t = orderable_f32(ot > 0x007fffffu + 2u ? ot - 2u : 0x007fffffu);       // (0x007fffff is -inf in that order)
    if (l == 0) tau[q] = (__popcll(good) >= K && kth) ? t : -INFINITY;
}

template <int D, int KIND>
__global__ __launch_bounds__(256) void k_tau_seed(int U, int n_local, const float *__restrict__ users_tab,
                                                  const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                  const float *__restrict__ sig_u, const float *__restrict__ sig_i,
                                                  float c_val, const float *__restrict__ c_dev,
                                                  const uint32_t *__restrict__ mask_bits, int item_offset, int K,
                                                  const int32_t *__restrict__ seed, float *__restrict__ tau,
                                                  const float *__restrict__ qpart = nullptr, int n_part = 0, uint32_t *__restrict__ qmax_bits = nullptr,
                                                  const int32_t *__restrict__ mask_ptr = nullptr, const int32_t *__restrict__ mask_idx = nullptr) {
    // (after k_eval_prologue_prep: one extra block turns its per-block item norms into max |q| for the listing pass)
    if (qpart && blockIdx.x == gridDim.x - 1) { qmax_from_parts(qpart, n_part, qmax_bits); return; }
    tau_seed_block<D, KIND>(blockIdx.x, U, n_local, users_tab, user_ids, items, sig_u, sig_i, c_val, c_dev, mask_bits, item_offset, K, seed, tau,
                            mask_ptr, mask_idx);
}

// Seeded ranking under the bf16 filter: the operand copies and the seeded thresholds do not depend on each other -- two
// short, latency-bound kernels -- and go out as ONE launch (blocks [0, n_prep) convert, the others score seeds).
template <int D, int KIND, bool HALF = false>
__global__ __launch_bounds__(256) void k_prep_tau_seed(int n_prep, int U, int n_local, const float *__restrict__ users_tab,
                                                       const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                       uint4 *__restrict__ users_bf, uint4 *__restrict__ items_bf,
                                                       float *__restrict__ unorm, uint32_t *__restrict__ qmax_bits,
                                                       const float *__restrict__ sig_u, const float *__restrict__ sig_i,
                                                       float c_val, const float *__restrict__ c_dev,
                                                       const uint32_t *__restrict__ mask_bits, int item_offset, int K,
                                                       const int32_t *__restrict__ seed, float *__restrict__ tau,
                                                       const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_idx) {
    if ((int)blockIdx.x < n_prep)
        bf16_prep_c_block<D, KIND, kPrepTrips, HALF>(blockIdx.x, U, n_local, users_tab, user_ids, items, sig_u, sig_i, c_dev ? *c_dev : c_val, users_bf,
                                                     items_bf, unorm, qmax_bits);
    else
        tau_seed_block<D, KIND>(blockIdx.x - n_prep, U, n_local, users_tab, user_ids, items, sig_u, sig_i, c_val, c_dev, mask_bits,
                                item_offset, K, seed, tau, mask_ptr, mask_idx);
}

// ----------------------------------------------------------------------------
// k_tau: one wave per user.  tau = the K-th largest of the user's n_splits*32 class maxima of pass 0
// (-inf while fewer than K classes saw an unmasked item: everything is listed then).
// MSB-first radix select on the order-preserving integer image of the floats; the state is one 64-bit
// lane mask per register (wave-uniform, SGPRs): per bit and register one v_and + v_cmp and a few scalar
// instructions, no cross-lane data movement.
// ----------------------------------------------------------------------------
constexpr int kSelWaves = 4;

// The one-wave-per-user kernels are bound by the CU's single scalar unit: their loops are unrolled over a
// COMPILE-TIME register count (no per-register guards) and the count is dispatched outside.
template <int NREG, bool REPAIR = false>
__global__ __launch_bounds__(64 * kSelWaves) void k_tau(int U, int n_splits, int K, const float *__restrict__ maxima,
                                                        const int32_t *__restrict__ blk_flag, float *__restrict__ tau,
                                                        const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits,
                                                        float c_val, const float *__restrict__ c_dev, int d_filter, int per_log2, int half,
                                                        const float *__restrict__ mscale) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * kSelWaves + wid;
    if (q >= U) return;
    // repair round: only the users of re-listed user blocks were sampled; their threshold is raised, never lowered
    if (REPAIR && blk_flag[q / kUsersPerBlock] == 0) return;
    // 2^per_log2 class maxima per (split, user): 32 (k_score_stream, k_score_sample_b) or 16 (k_score_sample_c)
    const int per = 1 << per_log2;
    const int n = n_splits * per;
    uint32_t key[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int e = j * 64 + lane;                              // split e / per, class e % per
        float m = -INFINITY;
        if (e < n) m = maxima[(((size_t)(e >> per_log2) * U + q) << per_log2) + (e & (per - 1))];
        key[j] = m > -INFINITY ? f32_orderable(m) : 0u;           // 0 = the class saw no unmasked item
    }
    int n_valid = 0;
    uint64_t cand[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) { cand[j] = __ballot(key[j] != 0u); n_valid += __popcll(cand[j]); }
    float t_out = -INFINITY;
    if (n_valid >= K) {
        int remaining = K, alive = n_valid;
        uint32_t prefix = 0u;                                     // bits decided so far of the K-th largest key
        int bit = 31;
        for (; bit >= 0 && alive > 1; --bit) {
            const uint32_t m = 1u << bit;
            uint64_t ones[NREG];
            int n1 = 0;
#pragma unroll
            for (int j = 0; j < NREG; ++j) { ones[j] = __ballot((key[j] & m) != 0u) & cand[j]; n1 += __popcll(ones[j]); }
            if (n1 >= remaining) {
                prefix |= m; alive = n1;
#pragma unroll
                for (int j = 0; j < NREG; ++j) cand[j] = ones[j];
            } else {
                remaining -= n1; alive -= n1;
#pragma unroll
                for (int j = 0; j < NREG; ++j) cand[j] &= ~ones[j];
            }
        }
        if (bit >= 0) {                                           // one candidate left: it is the K-th largest
#pragma unroll
            for (int j = 0; j < NREG; ++j)
                if (cand[j]) prefix = __shfl(key[j], __ffsll((long long)cand[j]) - 1, kWave);
        }
        t_out = orderable_f32(prefix);                            // (all 32 bits decided when equal maxima remain)
    }
    // maxima of bf16 scores (k_score_sample_b): K items score >= t_out - margin in fp32
    // (mscale: the fp16 filter states its margin on the accumulator; a score that leaves it times sig_u -- RUBI_BOTH -- is off by sig_u times that)
    if (unorm) t_out -= 1.01f * filter_margin_of(half, d_filter, unorm[q], __uint_as_float(*qmax_bits), c_dev ? *c_dev : c_val) * (mscale ? mscale[q] : 1.0f);
    if (lane == 0) tau[q] = REPAIR ? fmaxf(tau[q], t_out) : t_out;
}

// ----------------------------------------------------------------------------
// k_select: one wave per user.  Gathers the user's candidate lists of all splits, finds the K-th best
// key by the same radix select over 64-bit keys held in registers (kSelRegs per lane), sorts the K
// survivors and writes them as (score, id) rows to out_val/out_idx[0][u][:]; the other splits' rows are
// padded with (-inf, -1).
// ----------------------------------------------------------------------------

// Gather, select and sort for one user with NREG keys per lane (n <= 64*NREG); see k_select.
template <int NREG>
__device__ __forceinline__ void select_user(int q, int lane, int U, int n_splits, int n_out, int K, int cap, int n, int incl,
                                            const uint64_t *__restrict__ lists, uint64_t *s_top,
                                            float *__restrict__ out_val, int32_t *__restrict__ out_idx,
                                            int32_t *__restrict__ seed_out, int ql, int Ul) {
    // (ql, Ul): the query's index and the stride in the LISTS -- (q, U) except in the compact layout of a repair round
    // with seed_out the best kSeedWidth (>= K) candidates are ranked: the first K are the result, all of them the seeds
    // of the caller's next ranking (macr_score_topk seed_idx)
    const int Ksel = seed_out ? kSeedWidth : K;
    // element e of the gathered order lives in split s(e) at position e - offset(s); the addresses are resolved
    // first (uniform loop over the splits), then all loads are issued back to back
    size_t rel[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) rel[j] = (size_t)ql * cap + (j * 64 + lane);
    for (int s = 1; s < n_splits; ++s) {
        const int off = __builtin_amdgcn_readlane(incl, s - 1);                // exclusive prefix of split s
        if (off >= n) break;
        const size_t base = ((size_t)s * Ul + ql) * cap - off;
#pragma unroll
        for (int j = 0; j < NREG; ++j)
            if (j * 64 + lane >= off) rel[j] = base + (j * 64 + lane);
    }
    uint64_t key[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) key[j] = j * 64 + lane < n ? lists[rel[j]] : 0ull;
    // K-th largest of the n keys (distinct: ids differ): MSB-first radix select on lane masks
    uint64_t kth = 0ull;
    if (n >= Ksel) {
        uint64_t cand[NREG];
#pragma unroll
        for (int j = 0; j < NREG; ++j) cand[j] = __ballot(j * 64 + lane < n);
        int remaining = Ksel, alive = n;
        for (int bit = 63; bit >= 0 && alive > 1; --bit) {          // ends as soon as one candidate is left
            const uint32_t m = 1u << (bit & 31);
            uint64_t ones[NREG];
            int n1 = 0;
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const uint32_t word = bit >= 32 ? (uint32_t)(key[j] >> 32) : (uint32_t)key[j];
                ones[j] = __ballot((word & m) != 0u) & cand[j];
                n1 += __popcll(ones[j]);
            }
            if (n1 >= remaining) {
                alive = n1;
#pragma unroll
                for (int j = 0; j < NREG; ++j) cand[j] = ones[j];
            } else {
                remaining -= n1; alive -= n1;
#pragma unroll
                for (int j = 0; j < NREG; ++j) cand[j] &= ~ones[j];
            }
        }
        uint32_t hi = 0, lo = 0;
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
            if (cand[j]) {
                const int src_lane = __ffsll((long long)cand[j]) - 1;
                hi = __shfl((uint32_t)(key[j] >> 32), src_lane, kWave);
                lo = __shfl((uint32_t)key[j], src_lane, kWave);
            }
        }
        kth = ((uint64_t)hi << 32) | lo;
    }
    // survivors (exactly min(n, Ksel)) -> LDS by prefix popcount, then one wave-wide sort
    int base = 0;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const bool keep = key[j] != 0ull && key[j] >= kth;
        const uint64_t km = __ballot(keep);
        if (km) {
            if (keep) s_top[base + __popcll(km & ((1ull << lane) - 1ull))] = key[j];
            base += __popcll(km);
        }
    }
    const int kept = base;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint64_t k1 = lane < kept ? s_top[lane] : 0ull;
    k1 = wave_sort_desc(k1);
    if (lane < K) {
        out_val[(size_t)q * K + lane] = k1 ? key_score(k1) : -INFINITY;
        out_idx[(size_t)q * K + lane] = k1 ? key_id(k1) : -1;
        for (int s = 1; s < n_out; ++s) {
            out_val[((size_t)s * U + q) * K + lane] = -INFINITY;
            out_idx[((size_t)s * U + q) * K + lane] = -1;
        }
    }
    if (seed_out && lane < kSeedWidth) seed_out[(size_t)q * kSeedWidth + lane] = k1 ? key_id(k1) : -1;
}

template <bool REPAIR>
__global__ __launch_bounds__(64 * kSelWaves) void k_select(int U, int n_splits, int n_out, int K, int cap,
                                                           const uint64_t *__restrict__ lists,
                                                           const int32_t *__restrict__ counts, int32_t *overflow,
                                                           int ovf_per_user, const int32_t *__restrict__ run_if,
                                                           const int32_t *__restrict__ skip_blk,
                                                           float *__restrict__ out_val, int32_t *__restrict__ out_idx,
                                                           int32_t *__restrict__ seed_out) {
    __shared__ uint64_t s_top[kSelWaves][64];
    if (REPAIR && *run_if == 0) return;               // second selection: only after a repair round
    // ... and only for the user blocks that were listed again (skip_blk = blk_flag: 1 = in the repair round); the other
    // users' results stand -- after a bf16-filtered first round their lists hold bf16 scores, not ranking material
    const int blk_pos = skip_blk ? skip_blk[(blockIdx.x * kSelWaves) / kUsersPerBlock] : 0;       // repair round: position + 1 in ub_map
    if (REPAIR && skip_blk && blk_pos == 0) return;
    // first selection: a user block the listing pass gave up on (stale seeds) is ranked after the repair round only
    if (!REPAIR && skip_blk && skip_blk[(blockIdx.x * kSelWaves) / kUsersPerBlock] != 0) return;
    // wave-uniform values are made so explicitly (readfirstlane): the selection state then lives in SGPRs
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * kSelWaves + wid;
    if (q >= U) return;
    // list lengths of all splits at once (lane s <-> split s), exclusive prefix = offsets in the gathered order
    // where this query's lists are (repair_layout): as everywhere, or the compact layout of the re-listed query blocks
    int ql = q, Ul = U, nsl = n_splits;
    if (REPAIR && skip_blk) {
        const RepairLayout rl = repair_layout(n_splits, U, *run_if);
        if (rl.compact) { ql = (blk_pos - 1) * kUsersPerBlock + q % kUsersPerBlock; Ul = rl.stride; nsl = rl.slots; }
    }
    const int my_c = lane < nsl ? counts[(size_t)lane * Ul + ql] : 0;
    int incl = my_c;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, kWave); if (lane >= m) incl += o; }
    const int n_all = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, kWave));
    const int n = n_all < kSelRegs * 64 ? n_all : kSelRegs * 64;
    if (n_all > kSelRegs * 64 && lane == 0) overflow[ovf_per_user ? q : 0] = 1;
    // the common case (a few hundred candidates) runs the 4-keys-per-lane instance: these kernels are bound by the
    // CU's scalar unit and by registers, both proportional to the register count
    if (n <= 256) select_user<4>(q, lane, U, nsl, n_out, K, cap, n, incl, lists, s_top[wid], out_val, out_idx, seed_out, ql, Ul);
    else if (n <= 512) select_user<8>(q, lane, U, nsl, n_out, K, cap, n, incl, lists, s_top[wid], out_val, out_idx, seed_out, ql, Ul);
    else select_user<kSelRegs>(q, lane, U, nsl, n_out, K, cap, n, incl, lists, s_top[wid], out_val, out_idx, seed_out, ql, Ul);
}


// k_select_b: the selection after a bf16-filtered listing pass (see k_score_stream_b): the 64 best candidates by bf16
// score, the margin check, their exact fp32 scores, the ranking.  One wave per user.
template <int NREG>
__device__ __forceinline__ uint64_t gather_top64(int q, int lane, int U, int n_splits, int cap, int n, int incl,
                                                 const uint64_t *__restrict__ lists, uint64_t *s_top, bool sorted = true) {
    // candidate i of the query sits in the list of split s_i = #{s >= 1 : incl[s - 1] <= i} at position i - incl[s_i - 1]: the
    // splits are counted (one compare per split and register -- with 18 result slots per query the 64-bit address
    // selects this loop used to carry were a tenth of the kernel) and the offset fetched once, by lane
    int sj[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) sj[j] = 0;
    for (int s = 1; s < n_splits; ++s) {
        const int off = __builtin_amdgcn_readlane(incl, s - 1);
        if (off >= n) break;
#pragma unroll
        for (int j = 0; j < NREG; ++j) sj[j] += (j * 64 + lane >= off) ? 1 : 0;
    }
    uint64_t key[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const int before = __shfl(incl, sj[j] > 0 ? sj[j] - 1 : 0, kWave);       // (every lane takes part in the exchange)
        const size_t rel = ((size_t)sj[j] * U + q) * cap + (size_t)(j * 64 + lane - (sj[j] > 0 ? before : 0));
        key[j] = j * 64 + lane < n ? lists[rel] : 0ull;
    }
    uint64_t kth = 0ull;
    if (n > 64) {                                                   // the 64th largest of the n keys (distinct: ids differ)
        uint64_t cand[NREG];
#pragma unroll
        for (int j = 0; j < NREG; ++j) cand[j] = __ballot(j * 64 + lane < n);
        int remaining = 64, alive = n;
        for (int bit = 63; bit >= 0 && alive > 1; --bit) {
            const uint32_t m = 1u << (bit & 31);
            uint64_t ones[NREG];
            int n1 = 0;
#pragma unroll
            for (int j = 0; j < NREG; ++j) {
                const uint32_t word = bit >= 32 ? (uint32_t)(key[j] >> 32) : (uint32_t)key[j];
                ones[j] = __ballot((word & m) != 0u) & cand[j];
                n1 += __popcll(ones[j]);
            }
            if (n1 >= remaining) {
                alive = n1;
#pragma unroll
                for (int j = 0; j < NREG; ++j) cand[j] = ones[j];
            } else {
                remaining -= n1; alive -= n1;
#pragma unroll
                for (int j = 0; j < NREG; ++j) cand[j] &= ~ones[j];
            }
        }
        uint32_t hi = 0, lo = 0;
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
            if (cand[j]) {
                const int src_lane = __ffsll((long long)cand[j]) - 1;
                hi = __shfl((uint32_t)(key[j] >> 32), src_lane, kWave);
                lo = __shfl((uint32_t)key[j], src_lane, kWave);
            }
        }
        kth = ((uint64_t)hi << 32) | lo;
    }
    int base = 0;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const bool keep = key[j] != 0ull && key[j] >= kth;
        const uint64_t km = __ballot(keep);
        if (km) {
            if (keep) s_top[base + __popcll(km & ((1ull << lane) - 1ull))] = key[j];
            base += __popcll(km);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint64_t k1 = lane < base ? s_top[lane] : 0ull;
    return sorted ? wave_sort_desc(k1) : k1;                        // sorted: lane i = the (i+1)-th best candidate by bf16 score (0 = none)
}

// exact fp32 score of (query q, item id): the k-ascending fmaf chain and the epilogue of every other kernel here
template <int D, int KIND>
__device__ __forceinline__ uint64_t exact_key(int q, int id, const float *__restrict__ users_tab, const int32_t *__restrict__ user_ids,
                                              const float *__restrict__ items, const float *__restrict__ sig_u,
                                              const float *__restrict__ sig_i, float c, int item_offset) {
    const int it = id - item_offset;
    const float *ur = users_tab + (size_t)(user_ids ? user_ids[q] : q) * D, *ir = items + (size_t)it * D;
    float acc = 0.f;
#pragma unroll 8
    for (int k4 = 0; k4 < D / 4; ++k4) {
        const float4 a = ld4(ur + 4 * k4), b = ld4(ir + 4 * k4);
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    }
    float v = acc;
    if (score_uses_sig_i(KIND)) v = score_epilogue<KIND>(v, c, sig_i[it], score_uses_sig_u(KIND) ? sig_u[q] : 1.0f);
    return v == v ? make_key(v, id) : 0ull;                   // (a NaN score is no candidate: the fp32 listing test fails on it too)
}

// A query whose 64 best bf16 scores do not settle the top (k_select_b): every listed candidate gets its exact key, in
// place; the plain selection (select_user) then ranks the lists as if the fp32 listing pass had written them.
template <int NREG, int D, int KIND>
__device__ __forceinline__ void rescore_lists(int q, int ql, int lane, int U, int n_splits, int cap, int n, int incl, uint64_t *lists,
                                              const float *__restrict__ users_tab, const int32_t *__restrict__ user_ids,
                                              const float *__restrict__ items, const float *__restrict__ sig_u,
                                              const float *__restrict__ sig_i, float c, int item_offset) {
    size_t rel[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) rel[j] = (size_t)ql * cap + (j * 64 + lane);     // (ql, U: index and stride in the lists)
    for (int s = 1; s < n_splits; ++s) {
        const int off = __builtin_amdgcn_readlane(incl, s - 1);
        if (off >= n) break;
        const size_t base = ((size_t)s * U + ql) * cap - off;
#pragma unroll
        for (int j = 0; j < NREG; ++j)
            if (j * 64 + lane >= off) rel[j] = base + (j * 64 + lane);
    }
    for (int j = 0; j < NREG; ++j) {
        if (j * 64 + lane < n) {
            const uint64_t k = lists[rel[j]];
            lists[rel[j]] = k ? exact_key<D, KIND>(q, key_id(k), users_tab, user_ids, items, sig_u, sig_i, c, item_offset) : 0ull;
        }
    }
    // the selection reads these words again (through a pointer it was promised nobody writes): the stores first, and no load
    // of the compiler's moved above this point.  Workgroup scope: writer and reader are the same wave, whose CU's L1 is
    // write-through and coherent for its own workgroup; an agent-scope release here wrote back the XCD's whole L2 -- full of the
    // listing pass's lists -- in the wave that is the tail of the launch (profiles/r05_eval_fold_ab.txt on what that costs)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    asm volatile("" ::: "memory");
}

// (five waves per SIMD: the kernel is a chain of dependent memory trips -- list lengths, list entries, the candidates' rows --
// and at the 133 registers the 1024-candidate path asks for only three waves per SIMD were in flight: 36 -> 28 us seeded,
// 66 -> 49 us sampled on the Gowalla shape; eight waves per SIMD measured the same as five)
template <int D, int KIND, bool REPAIR = false>
__global__ __launch_bounds__(64 * kSelWaves, 5) void k_select_b(int U, int n_local, int n_splits, int n_out, int K, int cap,
                                                             uint64_t *lists, const int32_t *__restrict__ counts,
                                                             int32_t *overflow, int ovf_per_user, const int32_t *__restrict__ run_if,
                                                             const int32_t *__restrict__ skip_blk,
                                                             const float *__restrict__ users_tab, const int32_t *__restrict__ user_ids,
                                                             const float *__restrict__ items, const float *__restrict__ sig_u,
                                                             const float *__restrict__ sig_i, float c_val, const float *__restrict__ c_dev,
                                                             int item_offset, const float *__restrict__ unorm,
                                                             const uint32_t *__restrict__ qmax_bits,
                                                             float *__restrict__ out_val, int32_t *__restrict__ out_idx,
                                                             int32_t *__restrict__ seed_out, int half) {
    __shared__ uint64_t s_top[kSelWaves][64];
    // first selection: not the user blocks the listing pass gave up on; second one (REPAIR): only after a repair round,
    // and only the user blocks that were listed again (k_select)
    if (REPAIR && *run_if == 0) return;
    const int blk_pos = skip_blk ? skip_blk[(blockIdx.x * kSelWaves) / kUsersPerBlock] : 0;       // repair round: position + 1 in ub_map
    if (skip_blk && (blk_pos != 0) != REPAIR) return;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * kSelWaves + wid;
    if (q >= U) return;
    const float c = c_dev ? *c_dev : c_val;
    int ql = q, Ul = U, nsl = n_splits;               // where this query's lists are (repair_layout)
    if (REPAIR && skip_blk) {
        const RepairLayout rl = repair_layout(n_splits, U, *run_if);
        if (rl.compact) { ql = (blk_pos - 1) * kUsersPerBlock + q % kUsersPerBlock; Ul = rl.stride; nsl = rl.slots; }
    }
    const int my_c = lane < nsl ? counts[(size_t)lane * Ul + ql] : 0;
    int incl = my_c;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(incl, m, kWave); if (lane >= m) incl += o; }
    const int n_all = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, kWave));
    const int n = n_all < kSelRegs * 64 ? n_all : kSelRegs * 64;
    bool flag = n_all > kSelRegs * 64;
    uint64_t ka;
    if (n <= 64) {
        // at most 64 candidates (the usual case with seeded thresholds): all of them are scored exactly and sorted once
        // -- the bf16 order (a 64-lane bitonic sort) and the margin test would cost more than the extra fp32 scores
        ka = gather_top64<4>(ql, lane, Ul, nsl, cap, n, incl, lists, s_top[wid], false);
        if (flag && lane == 0) overflow[ovf_per_user ? q : 0] = 1;
        uint64_t ke = ka ? exact_key<D, KIND>(q, key_id(ka), users_tab, user_ids, items, sig_u, sig_i, c, item_offset) : 0ull;
        ke = wave_sort_desc(ke);
        if (lane < K) {
            out_val[(size_t)q * K + lane] = ke ? key_score(ke) : -INFINITY;
            out_idx[(size_t)q * K + lane] = ke ? key_id(ke) : -1;
            for (int s = 1; s < n_out; ++s) {
                out_val[((size_t)s * U + q) * K + lane] = -INFINITY;
                out_idx[((size_t)s * U + q) * K + lane] = -1;
            }
        }
        if (seed_out && lane < kSeedWidth) seed_out[(size_t)q * kSeedWidth + lane] = ke ? key_id(ke) : -1;
        return;
    }
    if (n <= 256) ka = gather_top64<4>(ql, lane, Ul, nsl, cap, n, incl, lists, s_top[wid]);
    else if (n <= 512) ka = gather_top64<8>(ql, lane, Ul, nsl, cap, n, incl, lists, s_top[wid]);
    else ka = gather_top64<kSelRegs>(ql, lane, Ul, nsl, cap, n, incl, lists, s_top[wid]);
    // no item outside these 64 may belong to the exact top K
    const int R = K;
    const uint32_t hi_r = __shfl((uint32_t)(ka >> 32), R - 1, kWave), hi_last = __shfl((uint32_t)(ka >> 32), 63, kWave);
    const float m2 = 2.02f * filter_margin_of(half, D, unorm[q], __uint_as_float(*qmax_bits), c) * ((half && KIND == MACR_SCORE_RUBI_BOTH) ? sig_u[q] : 1.0f);
    // (hi_r == 0: fewer than R candidates -- all of them matter)
    const float a_cut = hi_r ? orderable_f32(hi_r) - m2 : -INFINITY;
    if (flag && lane == 0) overflow[ovf_per_user ? q : 0] = 1;
    if (n > 64 && !(orderable_f32(hi_last) < a_cut)) {
        // more than 64 candidates inside two margins of the R-th (a flat top: scores that differ in the fifth digit):
        // every listed candidate is scored exactly and the plain selection ranks them -- a few queries per evaluation
        if (n <= 256) {
            rescore_lists<4, D, KIND>(q, ql, lane, Ul, nsl, cap, n, incl, lists, users_tab, user_ids, items, sig_u, sig_i, c, item_offset);
            select_user<4>(q, lane, U, nsl, n_out, K, cap, n, incl, lists, s_top[wid], out_val, out_idx, seed_out, ql, Ul);
        } else if (n <= 512) {
            rescore_lists<8, D, KIND>(q, ql, lane, Ul, nsl, cap, n, incl, lists, users_tab, user_ids, items, sig_u, sig_i, c, item_offset);
            select_user<8>(q, lane, U, nsl, n_out, K, cap, n, incl, lists, s_top[wid], out_val, out_idx, seed_out, ql, Ul);
        } else {
            rescore_lists<kSelRegs, D, KIND>(q, ql, lane, Ul, nsl, cap, n, incl, lists, users_tab, user_ids, items, sig_u, sig_i, c, item_offset);
            select_user<kSelRegs>(q, lane, U, nsl, n_out, K, cap, n, incl, lists, s_top[wid], out_val, out_idx, seed_out, ql, Ul);
        }
        return;
    }
    // exact scores of the candidates that can still belong to the top K (bf16 score within two margins of the K-th): the
    // arithmetic of k_tau_seed / the fp32 listing pass.  The others keep their bf16 key: each of them lies below every one
    // of the exact top K (its bf16 score + margin < K-th bf16 score - margin <= the K-th best exact score), so the first K
    // of the sorted keys are the exact ranking, and what follows only serves as seeds (any good candidates do).
    uint64_t ke = ka;
    if (ka && key_score(ka) >= a_cut) ke = exact_key<D, KIND>(q, key_id(ka), users_tab, user_ids, items, sig_u, sig_i, c, item_offset);
    ke = wave_sort_desc(ke);
    if (lane < K) {
        out_val[(size_t)q * K + lane] = ke ? key_score(ke) : -INFINITY;
        out_idx[(size_t)q * K + lane] = ke ? key_id(ke) : -1;
        for (int s = 1; s < n_out; ++s) {
            out_val[((size_t)s * U + q) * K + lane] = -INFINITY;
            out_idx[((size_t)s * U + q) * K + lane] = -1;
        }
    }
    if (seed_out && lane < kSeedWidth) seed_out[(size_t)q * kSeedWidth + lane] = ke ? key_id(ke) : -1;
    (void)n_local;
}

// ----------------------------------------------------------------------------
// k_repair_plan: what to do about candidate lists that overflowed.  A threshold can be far too loose for some users
// (seeds from a ranking the model has since moved away from; a sampling pass that missed the popular tiles): their
// lists were cut at `cap` entries in the listing pass, or held more than k_select ranks.  The entries that WERE
// listed are distinct unmasked items, and k_select has just written their top K: its K-th score is a lower bound of
// the user's K-th best score and, being rank K among >= cap listed items, a far tighter threshold.  One block per
// user block: flagged users get that threshold, a user block with any flagged user is appended to ub_map (the
// listing pass then runs again for those user blocks only) and the list lengths of its users are cleared.
// When the thresholds came from seeds, the re-listed user blocks are also SAMPLED first (the sampling pass and k_tau
// restricted to them): a first list that took everything holds the first `cap` items of its tile range, whose K-th
// best is at quantile K/cap of the catalogue -- on 41k items ~1600 candidates, more than k_select ranks -- while the
// sampled threshold sits at rank ~8K whatever happened before.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(kUsersPerBlock) void k_repair_plan(int U, int K, int n_slots, const int32_t *__restrict__ user_ovf,
                                                                const float *__restrict__ out_val, float *__restrict__ tau,
                                                                int32_t *__restrict__ counts, int32_t *__restrict__ ub_map,
                                                                int32_t *__restrict__ blk_flag, int32_t *n_ub, int32_t *n_done,
                                                                int32_t *stats) {
    const int ub = blockIdx.x, q = ub * kUsersPerBlock + threadIdx.x;
    const bool stopped = blk_flag[ub] != 0;               // the listing pass gave up on this user block (stale seeds)
    const bool f = q < U && (user_ovf[q] != 0 || stopped);
    if (f && !stopped) tau[q] = fmaxf(tau[q], out_val[(size_t)q * K + (K - 1)]);      // (a stopped block was not ranked)
    const bool any = __syncthreads_or(f) != 0;
    if (any) {
        if (q < U)
            for (int sl = 0; sl < n_slots; ++sl) counts[(size_t)sl * U + q] = 0;
        // blk_flag: position among the re-listed query blocks + 1 (the compact list layout of the repair round, repair_layout)
        if (threadIdx.x == 0) { const int pos = atomicAdd(n_ub, 1); ub_map[pos] = ub; blk_flag[ub] = pos + 1; }
    }
    // the last block to get here knows how many query blocks are listed again: if they take the compact layout, its
    // counts (which overlay other queries' counts, all read by now) start at zero
    // (one agent-scope release per block, behind the barrier that orders the block's writes before it: a release by every wave
    // writes back its XCD's L2 each time -- profiles/r05_eval_fold_ab.txt)
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); s_last = atomicAdd(n_done, 1) == (int)gridDim.x - 1; }
    __syncthreads();
    if (!s_last) return;
    const int n = __hip_atomic_load(n_ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (macr_score_topk_first_round: the caller learns here whether this round's result stands)
    if (stats && threadIdx.x == 0) { stats[0] = n; stats[1] = 0; }
    if (n == 0) return;
    const RepairLayout rl = repair_layout(n_slots, U, n);
    if (!rl.compact) return;
    const size_t words = (size_t)rl.slots * rl.stride;
    for (size_t k = threadIdx.x; k < words; k += kUsersPerBlock) counts[k] = 0;
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
                                                      const float *__restrict__ sig_i, float c_val,
                                                      const float *__restrict__ c_dev, float *__restrict__ out) {
    const float c = c_dev ? *c_dev : c_val;
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
    const float su = score_uses_sig_u(KIND) ? sig_u[q] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int id = it0 + il;
        if (id < n_local) {
            float v = acc[r];
            if (score_uses_sig_i(KIND)) v = score_epilogue<KIND>(v, c, sig_i[id], su);
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
// k_topk_scores_wide: the same for MACR_MAX_TOPK_FUSED < K <= MACR_MAX_TOPK per round (c_top_k_array_index has no bound on
// top_k, tools.h:13-22; the reference's CLIs use 20, its tuning scripts up to 100).  One wave per row; candidates above
// the running threshold collect in a 256-key LDS buffer; when it could overflow, the K-th largest is found by the MSB-first
// radix select on lane masks (four keys per lane) and the best K move to the front.  The final K are ordered by rank
// counting (every key against every other: K <= 128).  Not a hot path: correctness first.
// ----------------------------------------------------------------------------
constexpr int kWideCap = 256;
// keep the best K of keys[0..n) (n <= kWideCap, distinct keys) at the front; returns the K-th largest (0: n < K)
__device__ __forceinline__ uint64_t keep_best_wide(uint64_t *keys, int n, int K) {
    constexpr int NREG = kWideCap / 64;
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (n <= K) return 0ull;                               // (uniform)
    uint64_t key[NREG], cand[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        key[j] = j * 64 + lane < n ? keys[j * 64 + lane] : 0ull;
        cand[j] = __ballot(j * 64 + lane < n);
    }
    int remaining = K, alive = n;
    for (int bit = 63; bit >= 0 && alive > 1; --bit) {
        uint64_t ones[NREG];
        int n1 = 0;
#pragma unroll
        for (int j = 0; j < NREG; ++j) { ones[j] = __ballot((key[j] >> bit) & 1ull) & cand[j]; n1 += __popcll(ones[j]); }
        if (n1 >= remaining) {
            alive = n1;
#pragma unroll
            for (int j = 0; j < NREG; ++j) cand[j] = ones[j];
        } else {
            remaining -= n1; alive -= n1;
#pragma unroll
            for (int j = 0; j < NREG; ++j) cand[j] &= ~ones[j];
        }
    }
    uint64_t kth = 0ull;
#pragma unroll
    for (int j = 0; j < NREG; ++j)
        if (cand[j]) {                                     // (uniform) exactly one register holds the one candidate left
            const int src = __ffsll((long long)cand[j]) - 1;
            kth = ((uint64_t)__shfl((uint32_t)(key[j] >> 32), src, kWave) << 32) | __shfl((uint32_t)key[j], src, kWave);
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int base = 0;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const bool keep = key[j] != 0ull && key[j] >= kth;
        const uint64_t km = __ballot(keep);
        if (keep) keys[base + __popcll(km & ((1ull << lane) - 1ull))] = key[j];
        base += __popcll(km);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return kth;
}

// the final K (cnt <= K <= 128 keys at the front of `keys`) in order: rank = number of larger keys
__device__ __forceinline__ void emit_wide(const uint64_t *keys, int cnt, int K, int32_t *__restrict__ idx_row,
                                          float *__restrict__ val_row, int id_offset) {
    const int lane = threadIdx.x & 63;
    uint64_t mine[2]; int rank[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) mine[h] = h * 64 + lane < cnt ? keys[h * 64 + lane] : 0ull;
    for (int e = 0; e < cnt; ++e) {
        const uint64_t other = keys[e];
#pragma unroll
        for (int h = 0; h < 2; ++h) rank[h] += other > mine[h] ? 1 : 0;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (mine[h]) {
            idx_row[rank[h]] = key_id(mine[h]) + id_offset;
            if (val_row) val_row[rank[h]] = key_score(mine[h]);
        }
    for (int e = cnt + lane; e < K; e += kWave) {          // fewer than K candidates
        idx_row[e] = -1;
        if (val_row) val_row[e] = -INFINITY;
    }
}
// one round of admission: `key` per lane (0 = none) against the running threshold; the buffer is compacted to its best K
// when it could overflow
__device__ __forceinline__ void admit_wide(uint64_t *keys, int &cnt, uint64_t &thr_key, uint64_t key, int K) {
    const int lane = threadIdx.x & 63;
    bool cand = key > thr_key;
    uint64_t bal = __ballot(cand);
    if (bal == 0ull) return;
    if (cnt + __popcll(bal) > kWideCap) {
        const uint64_t kth = keep_best_wide(keys, cnt, K);
        if (kth) { thr_key = kth; cnt = K; }
        cand = key > thr_key;
        bal = __ballot(cand);
    }
    if (cand) keys[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = key;
    cnt += __popcll(bal);
}

// mask_bits (may be NULL): the (item tile, query) bitmap of macr_mask_bits_build with `mask_stride` queries; row r of the
// matrix is query q0 + r; a masked column is no candidate at all (as in the fused ranking).  Ids leave as column + id_offset.
// Rows may be ranked PAST 128 in rounds of <= 128 (macr_topk_scores with any K): a round writes positions out_off .. out_off + K - 1
// of rows `out_stride` long, admits only keys below the row's `below` key (the last key of the round before; 0 = the row ran out
// of candidates) and leaves its own last key in `last_key` -- the keys are a total order (score descending, id ascending), so the
// rounds concatenate to exactly the ranking one pass with a larger buffer would give.
__global__ __launch_bounds__(256) void k_topk_scores_wide(const float *__restrict__ scores, int cols, int rows, int K,
                                                          int32_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                          const uint32_t *__restrict__ mask_bits, int mask_stride, int q0,
                                                          int id_offset, int out_stride, int out_off,
                                                          const uint64_t *__restrict__ below, uint64_t *__restrict__ last_key) {
    __shared__ uint64_t s_keys[4][kWideCap];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    if (row >= rows) return;
    const float *src = scores + (size_t)row * cols;
    uint64_t *keys = s_keys[wid];
    int cnt = 0;
    uint64_t thr_key = 0ull;                        // admission: key > thr_key
    const bool bounded = below != nullptr;
    const uint64_t ceil_key = bounded ? below[row] : 0ull;
    for (int base = 0; base < cols; base += kWave) {
        const int cidx = base + lane;
        uint64_t key = cidx < cols ? make_key(src[cidx], cidx) : 0ull;
        if (mask_bits && cidx < cols && ((mask_bits[(size_t)(cidx >> 5) * mask_stride + q0 + row] >> (cidx & 31)) & 1u)) key = 0ull;
        if (bounded && key >= ceil_key) key = 0ull;               // ranked by an earlier round (ceil_key 0: nothing is left)
        admit_wide(keys, cnt, thr_key, key, K);
    }
    if (keep_best_wide(keys, cnt, K)) cnt = K;
    const size_t o = (size_t)row * out_stride + out_off;
    emit_wide(keys, cnt, K, out_idx + o, out_val ? out_val + o : nullptr, id_offset);
    if (last_key) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint64_t lo = ~0ull;
        for (int e = lane; e < cnt; e += kWave) lo = keys[e] < lo ? keys[e] : lo;
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const uint64_t other = ((uint64_t)__shfl_xor((uint32_t)(lo >> 32), sft, kWave) << 32) | __shfl_xor((uint32_t)lo, sft, kWave);
            lo = other < lo ? other : lo;
        }
        if (lane == 0) last_key[row] = cnt == K ? lo : 0ull;     // a short round: the row has no further candidates
    }
}

// k_topk_merge for MACR_MAX_TOPK_FUSED < K <= MACR_MAX_TOPK: the W*K entries stream through the same 256-key buffer.
__global__ __launch_bounds__(256) void k_topk_merge_wide(int W, int U, int K, const float *__restrict__ vals,
                                                         const int32_t *__restrict__ idxs,
                                                         const int32_t *__restrict__ fill_ptr,
                                                         const int32_t *__restrict__ fill_idx,
                                                         float *__restrict__ out_val, int32_t *__restrict__ out_idx,
                                                         int32_t *__restrict__ out_cnt) {
    __shared__ uint64_t s_keys[4][kWideCap];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wid;
    if (q >= U) return;
    uint64_t *keys = s_keys[wid];
    int cnt = 0;
    uint64_t thr_key = 0ull;
    for (int s = 0; s < W; ++s) {
        for (int base = 0; base < K; base += kWave) {
            const int k = base + lane;
            uint64_t key = 0ull;
            if (k < K) {
                const size_t o = ((size_t)s * U + q) * K + k;
                const int32_t id = idxs[o];
                if (id >= 0) key = make_key(vals[o], id);
            }
            admit_wide(keys, cnt, thr_key, key, K);
        }
    }
    if (keep_best_wide(keys, cnt, K)) cnt = K;
    emit_wide(keys, cnt, K, out_idx + (size_t)q * K, out_val + (size_t)q * K, 0);
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

// the item and the user branch factors of one evaluation in ONE launch (blocks [0, nb_a) are the first set)
template <int LPR>
__global__ __launch_bounds__(256) void k_branch_sigmoid2(int nb_a, const float *__restrict__ rows_a, const int32_t *__restrict__ idx_a, int n_a,
                                                         const float *__restrict__ w_a, float *__restrict__ out_a,
                                                         const float *__restrict__ rows_b, const int32_t *__restrict__ idx_b, int n_b,
                                                         const float *__restrict__ w_b, float *__restrict__ out_b) {
    constexpr int d = 4 * LPR;
    const bool first = (int)blockIdx.x < nb_a;
    const float *rows = first ? rows_a : rows_b, *w = first ? w_a : w_b;
    const int32_t *idx = first ? idx_a : idx_b;
    float *out = first ? out_a : out_b;
    const int n = first ? n_a : n_b, blk = first ? blockIdx.x : blockIdx.x - nb_a;
    const int sub = threadIdx.x % LPR;
    const int r = blk * (256 / LPR) + threadIdx.x / LPR;
    if (r >= n) return;
    const size_t src = idx ? (size_t)idx[r] : (size_t)r;
    const float s = group_sum<LPR>(dot4(ld4(rows + src * d + 4 * sub), ld4(w + 4 * sub)));       // (k_branch_sigmoid's arithmetic)
    if (sub == 0) out[r] = sigmoid_acc(s);
}

// The launches an evaluation starts with, as ONE (macr_score_topk_prologue): the branch factors of the items (blocks
// [0, nb_a)) and of the queries (the next nb_b blocks; nb_b may be 0) -- k_branch_sigmoid's arithmetic -- and, in the
// remaining blocks, k_topk_ws_init's statements on the head of the ranking workspace.
template <int LPR>
__global__ __launch_bounds__(256) void k_eval_prologue(int nb_a, const float *__restrict__ rows_a, int n_a, const float *__restrict__ w_a,
                                                       float *__restrict__ out_a, int nb_b, const float *__restrict__ rows_b,
                                                       const int32_t *__restrict__ idx_b, int n_b, const float *__restrict__ w_b,
                                                       float *__restrict__ out_b, uint32_t *__restrict__ base, size_t n_zero, size_t n_tau,
                                                       uint32_t tau_bits, int set_tau, size_t n_max) {
    constexpr int d = 4 * LPR;
    if ((int)blockIdx.x >= nb_a + nb_b) {
        const size_t blk = blockIdx.x - (nb_a + nb_b), nblk = gridDim.x - (nb_a + nb_b);
        const size_t total = n_zero + n_tau + n_max;
        for (size_t w = blk * (size_t)blockDim.x + threadIdx.x; w < total; w += nblk * blockDim.x) {
            if (w < n_zero) base[w] = 0u;
            else if (w < n_zero + n_tau) { if (set_tau) base[w] = tau_bits; }
            else base[w] = 0xffffffffu;
        }
        return;
    }
    const bool first = (int)blockIdx.x < nb_a;
    const float *rows = first ? rows_a : rows_b, *w = first ? w_a : w_b;
    const int32_t *idx = first ? nullptr : idx_b;
    float *out = first ? out_a : out_b;
    const int n = first ? n_a : n_b, blk = first ? blockIdx.x : blockIdx.x - nb_a;
    const int sub = threadIdx.x % LPR;
    const int r = blk * (256 / LPR) + threadIdx.x / LPR;
    if (r >= n) return;
    const size_t src = idx ? (size_t)idx[r] : (size_t)r;
    const float s = group_sum<LPR>(dot4(ld4(rows + src * d + 4 * sub), ld4(w + 4 * sub)));       // (k_branch_sigmoid's arithmetic)
    if (sub == 0) out[r] = sigmoid_acc(s);
}

// k_eval_prologue and the fp16 filter's operand copies as ONE launch (macr_score_topk_prologue_prep): the rows an evaluation
// starts by reading twice -- once for the branch factors, once for the copies, 15 MB at the Gowalla shape and COLD after a
// log interval of training (profiles/r06_eval_cold_start.txt) -- are read once.  Blocks [0, n_prep): bf16_prep_c_block<FUSED>
// (branch factors computed from the rows in hand, copies, |u|, per-block max |q|); the rest: the workspace head.
template <int D, int KIND>
__global__ __launch_bounds__(256) void k_eval_prologue_prep(int n_prep, int U, int n_local, const float *__restrict__ users_tab,
                                                            const int32_t *__restrict__ user_ids, const float *__restrict__ items,
                                                            const float *__restrict__ w_item, const float *__restrict__ w_user,
                                                            float *__restrict__ sig_i, float *__restrict__ sig_u,
                                                            float c_val, const float *__restrict__ c_dev,
                                                            uint4 *__restrict__ users_c, uint4 *__restrict__ items_c,
                                                            float *__restrict__ unorm, float *__restrict__ qpart,
                                                            uint32_t *__restrict__ base, size_t n_zero, size_t n_tau,
                                                            uint32_t tau_bits, int set_tau, size_t n_max) {
    if ((int)blockIdx.x >= n_prep) {
        const size_t blk = blockIdx.x - n_prep, nblk = gridDim.x - n_prep;
        const size_t total = n_zero + n_tau + n_max;
        for (size_t w = blk * (size_t)blockDim.x + threadIdx.x; w < total; w += nblk * blockDim.x) {
            if (w < n_zero) base[w] = 0u;
            else if (w < n_zero + n_tau) { if (set_tau) base[w] = tau_bits; }
            else base[w] = 0xffffffffu;
        }
        return;
    }
    bf16_prep_c_block<D, KIND, kPrepTripsAlone, true, true>(blockIdx.x, U, n_local, users_tab, user_ids, items, nullptr, nullptr,
                                                            c_dev ? *c_dev : c_val, users_c, items_c, unorm, nullptr,
                                                            w_item, w_user, sig_i, sig_u, qpart);
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

// 1 / log2(pos + 2) for pos = 0 .. 127 in float64 (the DCG discount of rank position pos: macr_mf/train.py:61-79, evaluate_foldout.h:71-88),
// tabulated: the double-precision log2 and division per lane were about half of k_metrics_mf's time (generated by Python: 1.0 / math.log2(pos + 2)).
__device__ const double kDcgDiscount[128] = {
    0x1.0000000000000p+0, 0x1.430939835353ep-1, 0x1.0000000000000p-1, 0x1.b903469050f73p-2,
    0x1.8c23246dc0aa0p-2, 0x1.6cc193acea9b5p-2, 0x1.5555555555555p-2, 0x1.430939835353ep-2,
    0x1.34413509f79ffp-2, 0x1.28009c1dd6454p-2, 0x1.1da3383416064p-2, 0x1.14b94f8d9641fp-2,
    0x1.0cf3ffed2d6acp-2, 0x1.0619dc46d3e15p-2, 0x1.0000000000000p-2, 0x1.f50b57eac5885p-3,
    0x1.eb22cc68aa6e3p-3, 0x1.e21e1180c5dabp-3, 0x1.d9dcd21439834p-3, 0x1.d244c78367a0dp-3,
    0x1.cb40589ac173ep-3, 0x1.c4bd95ba8d72bp-3, 0x1.bead76898f8cep-3, 0x1.b903469050f73p-3,
    0x1.b3b433f2eb070p-3, 0x1.aeb6f759c46fdp-3, 0x1.aa038eb0e3bfep-3, 0x1.a593062b38d8dp-3,
    0x1.a15f4c32b95a3p-3, 0x1.9d630dccc7ddfp-3, 0x1.999999999999ap-3, 0x1.95fec808a6094p-3,
    0x1.928ee7b0b4f23p-3, 0x1.8f46acf8c06e3p-3, 0x1.8c23246dc0aa0p-3, 0x1.8921a744e1aedp-3,
    0x1.863fd1a4a3053p-3, 0x1.837b7a642195ep-3, 0x1.80d2abffdfee9p-3, 0x1.7e439e8fed2b0p-3,
    0x1.7bccb2952736ep-3, 0x1.796c6c7b22305p-3, 0x1.772170b2747aap-3, 0x1.74ea804c2020fp-3,
    0x1.72c67602d3540p-3, 0x1.70b443a1f7c88p-3, 0x1.6eb2efbd2c1adp-3, 0x1.6cc193acea9b5p-3,
    0x1.6adf59c6e689dp-3, 0x1.690b7bca1f15ep-3, 0x1.67454177dda00p-3, 0x1.658bff53d6bf2p-3,
    0x1.63df15867d0dep-3, 0x1.623deedd496bap-3, 0x1.60a7ffe55458ap-3, 0x1.5f1cc61d1c5f1p-3,
    0x1.5d9bc73ac2288p-3, 0x1.5c2490845f2f3p-3, 0x1.5ab6b6386aaa4p-3, 0x1.5951d3046396fp-3,
    0x1.57f587883063fp-3, 0x1.56a179e4d652cp-3, 0x1.5555555555555p-3, 0x1.5410c9d09a12cp-3,
    0x1.52d38bb397b32p-3, 0x1.519d5372b6ce4p-3, 0x1.506ddd51defe8p-3, 0x1.4f44e92275a60p-3,
    0x1.4e223a06beda9p-3, 0x1.4d05963a1d8a7p-3, 0x1.4beec6ddbe115p-3, 0x1.4add97c942e46p-3,
    0x1.49d1d75f15f0ep-3, 0x1.48cb56640af5fp-3, 0x1.47c9e7da07ae1p-3, 0x1.46cd60dd6e321p-3,
    0x1.45d598850cb5ap-3, 0x1.44e267c45bba2p-3, 0x1.43f3a94fd9227p-3, 0x1.430939835353ep-3,
    0x1.4222f649fbc95p-3, 0x1.4140bf081c4a9p-3, 0x1.406274864d58fp-3, 0x1.3f87f8de0f744p-3,
    0x1.3eb12f67ab8e2p-3, 0x1.3dddfca9417e7p-3, 0x1.3d0e4646ed7c2p-3, 0x1.3c41f2f3ef9fbp-3,
    0x1.3b78ea64c23f4p-3, 0x1.3ab315420d981p-3, 0x1.39f05d1c68ad2p-3, 0x1.3930ac60d89e8p-3,
    0x1.3873ee4e00ec7p-3, 0x1.37ba0ee9f8387p-3, 0x1.3702faf8b610bp-3, 0x1.364e9ff30f403p-3,
    0x1.359cebfe36ed5p-3, 0x1.34edcde3bb953p-3, 0x1.34413509f79ffp-3, 0x1.3397116cededfp-3,
    0x1.32ef53978b4e7p-3, 0x1.3249ec9d46592p-3, 0x1.31a6ce14179f8p-3, 0x1.3105ea0ec499ep-3,
    0x1.3067331778205p-3, 0x1.2fca9c2aa39aap-3, 0x1.2f3018b2246ddp-3, 0x1.2e979c80a97bbp-3,
    0x1.2e011bcd54d68p-3, 0x1.2d6c8b2f960c4p-3, 0x1.2cd9df9b39ae5p-3, 0x1.2c490e5caaf3bp-3,
    0x1.2bba0d15648b9p-3, 0x1.2b2cd1b88de42p-3, 0x1.2aa15287c25cep-3, 0x1.2a1786100001cp-3,
    0x1.298f6326bb95ep-3, 0x1.2908e0e717dadp-3, 0x1.2883f6af3e1f6p-3, 0x1.28009c1dd6454p-3,
    0x1.277ec90f9c85ep-3, 0x1.26fe759d135d9p-3, 0x1.267f9a18501a1p-3, 0x1.26022f0ae0a4dp-3,
    0x1.25862d33c933dp-3, 0x1.250b8d8598a31p-3, 0x1.2492492492492p-3, 0x1.241a5964ec2e7p-3,
};

// The same, one WAVE per query (K <= 128): the K membership searches -- the dependent loads that are this kernel's time --
// run side by side (lane l owns rank positions l and 64 + l), the prefix recurrences are the thread version's own
// statements, run by every lane over the wave's hit mask (no memory in the loop; the discounts 1/log2(i+2) come from a table
// the block computes once), and lane l keeps the values of its positions.  A list shorter than K (ids < 0 from its first
// unused slot on) is completed with the query's masked ids in ascending order when a fill CSR is given -- what
// macr_topk_merge's fill does (-inf scores rank last, batch_test.py:124-134) without a launch of its own.
__global__ __launch_bounds__(256) void k_metrics_foldout_w(int U, int K, const int32_t *__restrict__ rankings,
                                                           const int32_t *__restrict__ fill_ptr, const int32_t *__restrict__ fill_idx,
                                                           const int32_t *__restrict__ gt_ptr, const int32_t *__restrict__ gt_idx,
                                                           float *__restrict__ results, int hr_in_ap_slot) {
    __shared__ double s_disc[128];
    if ((int)threadIdx.x < 128) s_disc[threadIdx.x] = kDcgDiscount[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= U) return;
    const int32_t *truth = gt_idx + gt_ptr[u];
    const int truth_len = gt_ptr[u + 1] - gt_ptr[u];
    int item[2];
    int n_valid = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = lane + 64 * h;
        item[h] = pos < K ? rankings[(size_t)u * K + pos] : -1;
        n_valid += __popcll(__ballot(item[h] >= 0));
    }
    if (fill_ptr && n_valid < K) {                             // wave-uniform
        const int f0 = fill_ptr[u], f1 = fill_ptr[u + 1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = lane + 64 * h, e = f0 + pos - n_valid;
            if (pos >= n_valid && pos < K && e < f1) item[h] = fill_idx[e];
        }
    }
    uint64_t hitm[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) hitm[h] = __ballot(item[h] >= 0 && in_sorted(truth, truth_len, item[h]));
    // Per position, by its own lane: hits so far (a popcount), precision, recall, reciprocal rank -- the double divisions run
    // once per lane, not once per lane and step.  The running sums whose float roundings depend on the order (sum_pre, dcg,
    // idcg) are accumulated by every lane over the hit mask with the thread version's statements; `pre` of a hit position comes
    // from that position's lane.
    float pre[2], rec[2], rr[2];
    int hits_at[2];
    const int first = hitm[0] ? __ffsll((long long)hitm[0]) - 1 : (hitm[1] ? 64 + __ffsll((long long)hitm[1]) - 1 : -1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = lane + 64 * h;
        const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
        hits_at[h] = h == 0 ? __popcll(hitm[0] & upto) : __popcll(hitm[0]) + __popcll(hitm[1] & upto);
        pre[h] = (float)(1.0 * hits_at[h] / (pos + 1));
        rec[h] = (float)(1.0 * hits_at[h] / truth_len);
        rr[h] = (first >= 0 && pos >= first) ? (float)(1.0 / (first + 1)) : 0.f;
    }
    float sum_pre = 0.f, dcg = 0.f, idcg = 0.f;
    float my_sum[2] = {0.f, 0.f}, my_dcg[2] = {0.f, 0.f}, my_idcg[2] = {0.f, 0.f};
    for (int i = 0; i < K; ++i) {
        const bool hit = (hitm[i >> 6] >> (i & 63)) & 1ull;
        const float pre_i = __shfl(i < 64 ? pre[0] : pre[1], i & 63, kWave);
        if (hit) {
            sum_pre += pre_i;
            dcg = (float)((double)dcg + s_disc[i]);
        }
        if (i < truth_len) idcg = (float)((double)idcg + s_disc[i]);
        if (i == lane) { my_sum[0] = sum_pre; my_dcg[0] = dcg; my_idcg[0] = idcg; }
        if (i == lane + 64) { my_sum[1] = sum_pre; my_dcg[1] = dcg; my_idcg[1] = idcg; }
    }
    float *res = results + (size_t)u * 5 * K;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = lane + 64 * h;
        if (pos < K) {
            res[0 * K + pos] = pre[h];
            res[1 * K + pos] = rec[h];
            res[2 * K + pos] = hr_in_ap_slot ? ((rec[h] != 0.f) ? 1.0f : 0.0f) : my_sum[h] / (float)truth_len;
            res[3 * K + pos] = my_dcg[h] / my_idcg[h];
            res[4 * K + pos] = rr[h];
        }
    }
}

// macr_mf/train.py:32-117 in float64: per query {precision, recall, ndcg, hit} x Ks.
struct KsArg { int32_t k[8]; int n; };
// One wave per query: lane i owns rank positions i and 64 + i (Kmax <= 128), so the membership searches and the log2 terms of a
// query run side by side; sums are fixed-shape shuffle trees (deterministic).
__global__ __launch_bounds__(256) void k_metrics_mf(int U, int Kmax, const int32_t *__restrict__ rankings,
                                                    const int32_t *__restrict__ cnt, const int32_t *__restrict__ gt_ptr,
                                                    const int32_t *__restrict__ gt_idx, KsArg Ks, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= U) return;
    const int32_t *truth = gt_idx + gt_ptr[u];
    const int truth_len = gt_ptr[u + 1] - gt_ptr[u];
    // lane l owns rank positions l and 64 + l (Kmax <= 128)
    bool hit[2]; double term[2];
    int n_valid = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = lane + 64 * h;
        const int item = pos < Kmax ? rankings[(size_t)u * Kmax + pos] : -1;
        n_valid += __popcll(__ballot(item >= 0));
        hit[h] = item >= 0 && in_sorted(truth, truth_len, item);
        term[h] = kDcgDiscount[pos];                              // DCG discount of position `pos` (pos < 128)
    }
    // cnt == NULL: a list's length is its number of ids >= 0 -- what macr_topk_merge reports for it
    const int len = cnt ? cnt[u] : n_valid;
    for (int qk = 0; qk < Ks.n; ++qk) {
        const int K = Ks.k[qk];
        const int m = len < K ? len : K;
        const int lim = truth_len < K ? truth_len : K;
        double hits = 0.0, dcg = 0.0, dcg_max = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                             // (positions 0..63 first, as before; then 64..127)
            const int pos = lane + 64 * h;
            const bool mine = hit[h] && pos < m;
            hits += (double)__popcll(__ballot(mine));
            dcg += wave_sum_d(mine ? term[h] : 0.0);
            dcg_max += wave_sum_d(pos < lim ? term[h] : 0.0);
        }
        if (lane == 0) {
            double *o = out + ((size_t)u * 4) * Ks.n;
            o[0 * Ks.n + qk] = m > 0 ? hits / m : NAN;
            o[1 * Ks.n + qk] = hits / truth_len;
            o[2 * Ks.n + qk] = dcg_max != 0 ? dcg / dcg_max : 0.0;
            o[3 * Ks.n + qk] = hits > 0 ? 1.0 : 0.0;
        }
    }
}

// k_metrics_mf AND the means over the query users (the "/ n_test_users" accumulation of macr_mf/train.py:286-290) in ONE
// launch: a block of 16 waves takes 64 consecutive queries, four per wave; a wave sums its queries' values in query order, the
// block the waves' sums in wave order (LDS) and writes one partial row; the block that takes the last ticket sums the partial
// rows -- 32 strided slices in block order, then the slices in order: the shape of the sum depends on (U, columns) only,
// never on who arrives last.  ticket: zero on entry, zero on return.
constexpr int kMeanWaves = 16, kMeanPerWave = 4, kMeanSlices = 32, kMeanCols = 32;
constexpr int kMeanPerBlock = kMeanWaves * kMeanPerWave;
__global__ __launch_bounds__(64 * kMeanWaves) void k_metrics_mf_mean(int U, int Kmax, const int32_t *__restrict__ rankings,
                                                                     const int32_t *__restrict__ cnt, const int32_t *__restrict__ gt_ptr,
                                                                     const int32_t *__restrict__ gt_idx, KsArg Ks, double *__restrict__ per_user,
                                                                     double *partials, int32_t *ticket, double *mean) {
    __shared__ double s_val[kMeanSlices][kMeanCols];      // [wave][column] first, [slice][column] in the last block
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ncols = 4 * Ks.n;
    if (lane < kMeanCols) s_val[wv][lane] = 0.0;          // (each wave touches its own row only: no barrier needed before the loop)
    double term[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) term[h] = kDcgDiscount[lane + 64 * h];                    // DCG discount of position lane + 64 h
    // A wave's four queries side by side in memory: one load for their five list bounds, then the four ranked lists and the
    // first 64 ids of the four truth lists in flight together -- two dependent trips for the wave instead of the 1 + log2(len)
    // of a binary search per query (what k_metrics_mf's 13 us are).  A truth list of <= 64 ids is searched across lanes.
    const int u0 = blockIdx.x * kMeanPerBlock + wv * kMeanPerWave;
    int bound = 0;
    if (lane <= kMeanPerWave) bound = gt_ptr[u0 + lane < U ? u0 + lane : U];
    int beg[kMeanPerWave], tlen[kMeanPerWave], item[kMeanPerWave][2], tr[kMeanPerWave], len_in[kMeanPerWave];
#pragma unroll
    for (int k = 0; k < kMeanPerWave; ++k) {
        beg[k] = __builtin_amdgcn_readlane(bound, k);
        tlen[k] = __builtin_amdgcn_readlane(bound, k + 1) - beg[k];
    }
#pragma unroll
    for (int k = 0; k < kMeanPerWave; ++k) {
        const int u = u0 + k;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = lane + 64 * h;
            item[k][h] = (u < U && pos < Kmax) ? rankings[(size_t)u * Kmax + pos] : -1;
        }
        tr[k] = (u < U && lane < tlen[k]) ? gt_idx[beg[k] + lane] : -1;
        len_in[k] = (cnt && u < U) ? cnt[u] : 0;
    }
#pragma unroll
    for (int k = 0; k < kMeanPerWave; ++k) {
        const int u = u0 + k;
        if (u >= U) break;                                // wave-uniform
        const int truth_len = tlen[k];
        bool hit[2];
        int n_valid = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) n_valid += __popcll(__ballot(item[k][h] >= 0));
        if (truth_len <= 64) {
            hit[0] = hit[1] = false;
            for (int j = 0; j < truth_len; ++j) {         // (wave-uniform bound; ids are >= 0, so an unused slot's -1 never matches)
                const int t = __builtin_amdgcn_readlane(tr[k], j);
                hit[0] = hit[0] || item[k][0] == t;
                hit[1] = hit[1] || item[k][1] == t;
            }
        } else {
            const int32_t *truth = gt_idx + beg[k];
#pragma unroll
            for (int h = 0; h < 2; ++h) hit[h] = item[k][h] >= 0 && in_sorted(truth, truth_len, item[k][h]);
        }
        const int len = cnt ? len_in[k] : n_valid;
        for (int qk = 0; qk < Ks.n; ++qk) {               // (k_metrics_mf's statements)
            const int K = Ks.k[qk];
            const int m = len < K ? len : K;
            const int lim = truth_len < K ? truth_len : K;
            double hits = 0.0, dcg = 0.0, dcg_max = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pos = lane + 64 * h;
                const bool mine = hit[h] && pos < m;
                hits += (double)__popcll(__ballot(mine));
                dcg += wave_sum_d(mine ? term[h] : 0.0);
                dcg_max += wave_sum_d(pos < lim ? term[h] : 0.0);
            }
            if (lane == 0) {
                const double v0 = m > 0 ? hits / m : NAN, v1 = hits / truth_len, v2 = dcg_max != 0 ? dcg / dcg_max : 0.0,
                             v3 = hits > 0 ? 1.0 : 0.0;
                s_val[wv][0 * Ks.n + qk] += v0; s_val[wv][1 * Ks.n + qk] += v1;
                s_val[wv][2 * Ks.n + qk] += v2; s_val[wv][3 * Ks.n + qk] += v3;
                if (per_user) {
                    double *o = per_user + ((size_t)u * 4) * Ks.n;
                    o[0 * Ks.n + qk] = v0; o[1 * Ks.n + qk] = v1; o[2 * Ks.n + qk] = v2; o[3 * Ks.n + qk] = v3;
                }
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < ncols) {
        double s = 0.0;
        for (int w = 0; w < kMeanWaves; ++w) s += s_val[w][threadIdx.x];       // (a wave without queries holds zeros)
        // write-through stores and loads (spmm_kernels.hip's publish / last_of): a partial row is out in memory before its
        // block's arrival is counted -- no fence: an agent-scope release by every wave writes back its XCD's L2 each time
        // (this launch took 238 us that way)
        __hip_atomic_store(partials + (size_t)blockIdx.x * ncols + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    const int col = threadIdx.x % kMeanCols, slice = threadIdx.x / kMeanCols;       // 1024 threads = 32 slices x 32 columns
    double acc = 0.0;
    if (col < ncols) {
        for (int b0 = slice; b0 < (int)gridDim.x; b0 += 8 * kMeanSlices) {          // eight loads in flight, added in block order
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = b0 + j * kMeanSlices;
                v[j] = b < (int)gridDim.x ? __hip_atomic_load(partials + (size_t)b * ncols + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    s_val[slice][col] = acc;
    __syncthreads();
    if ((int)threadIdx.x < ncols) {
        double t = 0.0;
        for (int k = 0; k < kMeanSlices; ++k) t += s_val[k][threadIdx.x];
        mean[threadIdx.x] = t / (double)U;
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // for the next launch
}

// column means in float64, one 1024-thread block per column, fixed-shape tree => deterministic
template <typename T>
__global__ __launch_bounds__(1024) void k_colmean(const T *__restrict__ in, int rows, int cols, double *__restrict__ out) {
    __shared__ double red[16];
    const int cidx = blockIdx.x;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;                  // four independent chains: loads stay in flight
    int r = threadIdx.x;
    for (; r + 3 * 1024 < rows; r += 4 * 1024) {
        s0 += (double)in[(size_t)r * cols + cidx];
        s1 += (double)in[(size_t)(r + 1024) * cols + cidx];
        s2 += (double)in[(size_t)(r + 2048) * cols + cidx];
        s3 += (double)in[(size_t)(r + 3072) * cols + cidx];
    }
    for (; r < rows; r += 1024) s0 += (double)in[(size_t)r * cols + cidx];
    double s = wave_sum_d((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < 16; ++k) t += red[k];
        out[cidx] = t / (double)rows;
    }
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

namespace macr {
// Launch geometry of the two streaming passes (see k_score_stream): grid sizes and the number of result slots a
// user block can have (= blocks overlapping its tile range).
struct StreamGeo { int ublocks, grid0, slots0, grid1, slots1, range1, grid_h; };   // range1: longest tile range of pass 1; grid_h: k_score_stream_h's blocks
static StreamGeo stream_geo(int U, int n_local, int d) {
    StreamGeo g;
    g.ublocks = (U + kUsersPerBlock - 1) / kUsersPerBlock;
    const int resident = d <= 64 ? 512 : 256;                 // 8-wave blocks resident on 256 CUs
    const int T1 = (n_local + kTileItems - 1) / kTileItems, sl = sample_log2(n_local), T0 = (T1 + (1 << sl) - 1) >> sl;
    auto plan = [&](int T, int &grid, int &slots) {
        const long long W = (long long)g.ublocks * T;
        long long G = resident;
        if (G > W / 8) G = W / 8;                             // at least eight tiles per block (prologue, result slots)
        if (G > (long long)g.ublocks * 60) G = (long long)g.ublocks * 60;   // at most ~62 blocks per user block
        if (G < 1) G = 1;
        const long long chunk = W / G;                        // shortest range
        grid = (int)G;
        slots = (int)((T + chunk - 1) / chunk + 1);
    };
    plan(T0, g.grid0, g.slots0);
    plan(T1, g.grid1, g.slots1);
    g.range1 = (int)(((long long)g.ublocks * T1 + g.grid1 - 1) / g.grid1);
    g.grid_h = g.grid1;
    if (kHTiles == 1) {                                       // 4-wave blocks: twice as many of them resident, shorter ranges, more result slots
        const long long W = (long long)g.ublocks * T1;
        long long G = 2LL * resident;
        if (G > W / 8) G = W / 8;
        if (G > (long long)g.ublocks * 60) G = (long long)g.ublocks * 60;
        if (G < 1) G = 1;
        const long long chunk = W / G;
        g.grid_h = (int)G;
        g.slots1 = std::max(g.slots1, (int)((T1 + chunk - 1) / chunk + 1));
    }
    return g;
}

// Workspace of macr_score_topk: counts[S1][U] | flags | shared_thr[U] (fallback kernel) | user_ovf[U] | ub_map[ublocks] |
// tau[U] | maxima[S0][U][32] |
// mask_bits[tiles][U] | lists[S1][U][cap]
struct TopkWs {
    float *tau, *maxima; int32_t *counts; int32_t *overflow, *user_ovf, *ub_map, *blk_flag; uint32_t *shared_thr, *mask_bits; uint64_t *lists;
    uint4 *users_bf, *items_bf; float *unorm;        // bf16 filter: operand copies, |u| per query user (max |q|: overflow[8], zeroed per call)
    uint4 *users_c, *items_c;
    float *qpart;
    int cap; size_t header_bytes, maxima_bytes, mask_bytes, lists_bytes, bytes;
};
static TopkWs carve_topk_ws(void *base, int U, int n_local, const StreamGeo &g, int d = 0) {
    TopkWs w;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { void *r = p ? p + off : nullptr; off += align_up(bytes, 256); return r; };
    // One list per (slot, user).  ~8K listed items per user in total, but item ids often follow popularity, so one
    // slot may receive nearly all of them: capacity is per slot, not divided by their number.
    w.cap = g.slots1 <= 2 ? 1024 : 512;
    w.counts = static_cast<int32_t *>(take((size_t)g.slots1 * U * 4));
    w.overflow = static_cast<int32_t *>(take(256));
    w.shared_thr = static_cast<uint32_t *>(take((size_t)U * 4));
    w.user_ovf = static_cast<int32_t *>(take((size_t)U * 4));           // per-user "list overflowed" flags of the first round
    w.ub_map = static_cast<int32_t *>(take((size_t)g.ublocks * 4));     // user blocks of the repair round (overflow[1] of them)
    w.blk_flag = static_cast<int32_t *>(take((size_t)g.ublocks * 4));   // 1 = this user block is in ub_map
    w.header_bytes = off;                                   // zeroed at the start of every call
    w.tau = static_cast<float *>(take((size_t)U * 4));
    w.maxima_bytes = (size_t)g.slots0 * U * 32 * 4;         // set to NaN (0xff bytes) at the start of every call
    w.maxima = static_cast<float *>(take(w.maxima_bytes));
    w.mask_bytes = (size_t)(n_tiles(n_local) + n_windows(n_local)) * U * 4;
    w.mask_bits = static_cast<uint32_t *>(take(w.mask_bytes));
    // (the wide ranking, K > MACR_MAX_TOPK_FUSED, keeps a block of dense score rows here: room for at least min(U, 32) of them)
    w.lists_bytes = std::max((size_t)g.slots1 * U * w.cap * 8, (size_t)(U < 32 ? U : 32) * n_local * 4);
    w.lists = static_cast<uint64_t *>(take(w.lists_bytes));
    w.users_bf = static_cast<uint4 *>(take((size_t)U * d * 4));          // (hi[d], lo[d]) bf16 per row
    w.items_bf = static_cast<uint4 *>(take((size_t)n_local * d * 4));
    w.unorm = static_cast<float *>(take((size_t)U * 4));
    w.users_c = static_cast<uint4 *>(take((size_t)U * d * 4));           // k_score_stream_c's copies (the epilogue in the operands)
    w.items_c = static_cast<uint4 *>(take(d ? items_c_bytes(n_local, d) : 0));
    w.qpart = static_cast<float *>(take(d ? (size_t)bf16_prep_c_blocks(U, n_local, d, kPrepTripsAlone) * 4 : 0));   // k_eval_prologue_prep's per-block item norms
    w.bytes = off;
    return w;
}
}  // namespace macr

// (-DMACR_EVAL_KERNELS_ONLY: the kernel templates and launch geometry alone, for a stand-alone kernel bench that #includes
// this file -- seconds to compile instead of minutes)
#ifndef MACR_EVAL_KERNELS_ONLY
extern "C" size_t macr_score_topk_workspace_bytes(int U, int n_local, int d) {
    if (U <= 0 || n_local <= 0 || !dim_supported(d)) return 0;
    return carve_topk_ws(nullptr, U, n_local, stream_geo(U, n_local, d), d).bytes;
}

// `filter` argument of the ranking entry points: MACR_EVAL_FILTER_ENV (0) follows MACR_EVAL_FILTER in the environment (f32 when
// unset), MACR_EVAL_FILTER_F32 (1), MACR_EVAL_FILTER_BF16 (2).  Per call: the library keeps no filter state (abi 10).
static int eval_filter_resolved(int filter) {            // MACR_EVAL_FILTER_F32 | _BF16 | _F16
    if (filter) return filter;
    const char *e = getenv("MACR_EVAL_FILTER");
    if (e && (e[0] == 'b' || e[0] == 'B')) return MACR_EVAL_FILTER_BF16;
    if (e && (e[0] == 'f' || e[0] == 'F') && e[1] == '1') return MACR_EVAL_FILTER_F16;      // "f16" ("f32": the default)
    return MACR_EVAL_FILTER_F32;
}
// the reduced-precision candidate filters share one path (operand copies, thresholds less a margin, k_select_b); `half`
// picks the fp16 kernels and margin on it.  (The c sweep has bf16 kernels only: the fp16 filter takes those there.)
static bool eval_filter_bf16(int filter) { return eval_filter_resolved(filter) != MACR_EVAL_FILTER_F32; }
static bool eval_filter_half(int filter) { return eval_filter_resolved(filter) == MACR_EVAL_FILTER_F16; }

extern "C" int macr_score_topk_uses_seeds(int U, int n_local, int d) {
    if (U <= 0 || n_local <= 0 || !dim_supported(d)) return 0;
    const StreamGeo geo = stream_geo(U, n_local, d);
    const bool list_all = n_local <= kSelRegs * 64 && geo.range1 * kTileItems <= carve_topk_ws(nullptr, U, n_local, geo).cap;
    return list_all ? 0 : 1;             // a shard this small lists every unmasked item: no thresholds, nothing to seed
}

extern "C" size_t macr_mask_bits_bytes(int U, int n_local) {
    if (U <= 0 || n_local <= 0) return 0;
    return (size_t)(n_tiles(n_local) + n_windows(n_local)) * U * 4;      // per-tile words, then per-window words (sampling pass)
}

extern "C" int macr_mask_bits_build(int U, int n_local, const int32_t *mask_ptr, const int32_t *mask_idx, int item_offset,
                                    uint32_t *mask_bits, void *stream) {
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "mask_bits_build: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(mask_ptr && mask_idx && mask_bits, MACR_E_INVALID, "mask_bits_build: null pointer");
    hipStream_t st = as_stream(stream);
    fill_words(mask_bits, macr_mask_bits_bytes(U, n_local) / 4, 0u, st);
    k_mask_bits<<<(U + 3) / 4, 256, 0, st>>>(U, n_local, mask_ptr, mask_idx, item_offset, mask_bits, n_tiles(n_local),
                                             sample_log2(n_local));
    MACR_CHECK_LAUNCH("mask_bits", st);
    return MACR_OK;
}

extern "C" int macr_score_topk_splits(int U, int n_local, int d) {
    // The streaming passes balance their own grid (stream_geo) and leave the merged result in list 0; more output
    // lists would only add padding for macr_topk_merge to read.  (n_splits > 1 still shapes the fallback kernel.)
    (void)U; (void)n_local; (void)d;
    return 1;
}

namespace macr {
template <bool REPAIR>
static void launch_k_tau(int tau_regs, int blocks, hipStream_t st, int U, int slots0, int K, const float *maxima,
                         const int32_t *blk_flag, float *tau, const float *unorm = nullptr, const uint32_t *qmax_bits = nullptr,
                         float c = 0.f, const float *c_dev = nullptr, int d_filter = 0, int per_log2 = 5, int half = 0,
                         const float *mscale = nullptr) {
    const int th = 64 * kSelWaves;
    // (unorm != NULL: the maxima are bf16 scores, tau = K-th largest - the filter's margin)
    if (tau_regs <= 1) k_tau<1, REPAIR><<<blocks, th, 0, st>>>(U, slots0, K, maxima, blk_flag, tau, unorm, qmax_bits, c, c_dev, d_filter, per_log2, half, mscale);
    else if (tau_regs <= 2) k_tau<2, REPAIR><<<blocks, th, 0, st>>>(U, slots0, K, maxima, blk_flag, tau, unorm, qmax_bits, c, c_dev, d_filter, per_log2, half, mscale);
    else if (tau_regs <= 4) k_tau<4, REPAIR><<<blocks, th, 0, st>>>(U, slots0, K, maxima, blk_flag, tau, unorm, qmax_bits, c, c_dev, d_filter, per_log2, half, mscale);
    else if (tau_regs <= 8) k_tau<8, REPAIR><<<blocks, th, 0, st>>>(U, slots0, K, maxima, blk_flag, tau, unorm, qmax_bits, c, c_dev, d_filter, per_log2, half, mscale);
    else if (tau_regs <= 16) k_tau<16, REPAIR><<<blocks, th, 0, st>>>(U, slots0, K, maxima, blk_flag, tau, unorm, qmax_bits, c, c_dev, d_filter, per_log2, half, mscale);
    else k_tau<32, REPAIR><<<blocks, th, 0, st>>>(U, slots0, K, maxima, blk_flag, tau, unorm, qmax_bits, c, c_dev, d_filter, per_log2, half, mscale);
}
}  // namespace macr

// (-DMACR_DEV_FAST: d = 64 and the kinds NORMAL / RUBI_BOTH only -- a build for kernel work that compiles in a fraction of the
// four minutes the full set of instantiations takes; never shipped: the Makefile does not know it)
#ifdef MACR_DEV_FAST
#define MACR_DISPATCH_K(Dv, kind, ...)                                                                    \
    switch (kind) {                                                                                       \
        case MACR_SCORE_NORMAL:            { constexpr int D = Dv, KIND = MACR_SCORE_NORMAL; __VA_ARGS__; } break;            \
        case MACR_SCORE_RUBI_BOTH:         { constexpr int D = Dv, KIND = MACR_SCORE_RUBI_BOTH; __VA_ARGS__; } break;         \
    }
#else
#define MACR_DISPATCH_K(Dv, kind, ...)                                                                    \
    switch (kind) {                                                                                       \
        case MACR_SCORE_NORMAL:            { constexpr int D = Dv, KIND = MACR_SCORE_NORMAL; __VA_ARGS__; } break;            \
        case MACR_SCORE_RUBI_BOTH:         { constexpr int D = Dv, KIND = MACR_SCORE_RUBI_BOTH; __VA_ARGS__; } break;         \
        case MACR_SCORE_RUBI:              { constexpr int D = Dv, KIND = MACR_SCORE_RUBI; __VA_ARGS__; } break;              \
        case MACR_SCORE_DIRECT_MINUS:      { constexpr int D = Dv, KIND = MACR_SCORE_DIRECT_MINUS; __VA_ARGS__; } break;      \
        case MACR_SCORE_DIRECT_MINUS_BOTH: { constexpr int D = Dv, KIND = MACR_SCORE_DIRECT_MINUS_BOTH; __VA_ARGS__; } break; \
    }
#endif
// what k_topk_ws_init / k_eval_prologue write to the head of the ranking workspace before a first round: counts, flags,
// shared_thr <- 0; tau <- -inf on the list-everything path; the class maxima <- NaN ("this class saw nothing") -- none for a
// seeded first round under the bf16 filter (no sampling pass: its repair round, if one follows, fills them itself)
struct WsHeadPlan { size_t n_zero, n_tau, n_max; int set_tau; unsigned grid; };
static WsHeadPlan ws_head_plan(const TopkWs &ws, const StreamGeo &geo, int U, int n_local, int K, int filter, bool seeded_first_round) {
    const bool list_all = n_local <= kSelRegs * 64 && geo.range1 * kTileItems <= ws.cap;
    const bool filter_bf16 = !list_all && eval_filter_bf16(filter);
    const bool bf16_merge = filter_bf16 && geo.slots0 * 16 >= 4 * K;
    const size_t n_max_words = list_all ? 0 : filter_bf16 ? (size_t)geo.slots0 * U * (bf16_merge ? 16 : 32) : ws.maxima_bytes / 4;
    WsHeadPlan p;
    p.n_zero = ws.header_bytes / 4;
    p.n_tau = (reinterpret_cast<char *>(ws.maxima) - reinterpret_cast<char *>(ws.tau)) / 4;
    p.n_max = (seeded_first_round && filter_bf16) ? 0 : n_max_words;
    p.set_tau = list_all ? 1 : 0;
    const size_t total = p.n_zero + p.n_tau + p.n_max, blocks = (total + 256 * 8 - 1) / (256 * 8);
    p.grid = (unsigned)(blocks < 1 ? 1 : blocks < 2048 ? blocks : 2048);
    return p;
}

#ifdef MACR_DEV_FAST
#define MACR_DISPATCH_DK(d, kind, ...)                              \
    switch (d) {                                                    \
        case 64:  MACR_DISPATCH_K(64, kind, __VA_ARGS__); break;    \
    }
#else
#define MACR_DISPATCH_DK(d, kind, ...)                              \
    switch (d) {                                                    \
        case 32:  MACR_DISPATCH_K(32, kind, __VA_ARGS__); break;    \
        case 64:  MACR_DISPATCH_K(64, kind, __VA_ARGS__); break;    \
        case 128: MACR_DISPATCH_K(128, kind, __VA_ARGS__); break;   \
        case 256: MACR_DISPATCH_K(256, kind, __VA_ARGS__); break;   \
    }
#endif

static inline bool score_kind_valid(int k) { return k >= MACR_SCORE_NORMAL && k <= MACR_SCORE_DIRECT_MINUS_BOTH; }

// mode 1 (macr_score_topk_first_round): the first round alone -- no repair round, no fallback kernel; stats tell whether
// its result stands.  mode 2 (macr_score_topk_repair_round): what the complete call launches after the first round, on the
// workspace a first-round call with the same arguments left behind.  mode 0: both.
static int score_topk_impl(int mode, int filter, int score_kind, int U, int n_local, int d, const float *users_tab,
                           const int32_t *user_ids, const float *items, const float *sig_u,
                           const float *sig_i, float c, const float *c_dev, const int32_t *mask_ptr, const int32_t *mask_idx,
                           const uint32_t *mask_bits_in, int item_offset, int K, int n_splits, const int32_t *seed_idx,
                           int32_t *seed_out, float *out_val, int32_t *out_idx, int32_t *stats, void *workspace,
                           size_t workspace_bytes, void *stream) {
    // MACR_TOPK_FALLBACK=1 in the environment runs the fallback kernel unconditionally (tests of that path)
    static const bool force_fallback = getenv("MACR_TOPK_FALLBACK") && getenv("MACR_TOPK_FALLBACK")[0] == '1';
    MACR_REQUIRE(score_kind_valid(score_kind), MACR_E_INVALID, "score_topk: score_kind=%d", score_kind);
    // MACR_EVAL_WS_READY: macr_score_topk_prologue has initialised the workspace head for exactly this call
    const bool ws_ready = (filter & MACR_EVAL_WS_READY) != 0;
    // MACR_EVAL_PREP_READY: macr_score_topk_prologue_prep has also written the fp16 filter's operand copies for this call
    const bool prep_ready = (filter & MACR_EVAL_PREP_READY) != 0;
    filter &= ~(MACR_EVAL_WS_READY | MACR_EVAL_PREP_READY);
    MACR_REQUIRE(!prep_ready || (ws_ready && eval_filter_resolved(filter) == MACR_EVAL_FILTER_F16), MACR_E_INVALID,
                 "score_topk: MACR_EVAL_PREP_READY goes with MACR_EVAL_WS_READY and the fp16 filter");
    MACR_REQUIRE(filter >= MACR_EVAL_FILTER_ENV && filter <= MACR_EVAL_FILTER_F16, MACR_E_INVALID, "score_topk: filter=%d", filter);
    MACR_REQUIRE(!ws_ready || mode != 2, MACR_E_INVALID, "score_topk: MACR_EVAL_WS_READY on a repair round");
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_topk: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_topk: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "score_topk: K=%d outside [1,%d]", K, MACR_MAX_TOPK);
    MACR_REQUIRE(users_tab && items && out_val && out_idx, MACR_E_INVALID, "score_topk: null pointer");
    MACR_REQUIRE((!score_uses_sig_i(score_kind) || sig_i) && (!score_uses_sig_u(score_kind) || sig_u), MACR_E_INVALID,
                 "score_topk: score_kind %d needs sig_i%s", score_kind, score_uses_sig_u(score_kind) ? " and sig_u" : "");
    MACR_REQUIRE((mask_ptr == nullptr) == (mask_idx == nullptr) || mask_ptr, MACR_E_INVALID, "score_topk: mask_idx without mask_ptr");
    const bool first_only = mode == 1, repair_only = mode == 2;
    if (n_splits <= 0) n_splits = macr_score_topk_splits(U, n_local, d);
    const int ublocks = (U + kUsersPerBlock - 1) / kUsersPerBlock;
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(n_splits <= 64, MACR_E_INVALID, "score_topk: n_splits=%d > 64", n_splits);
    MACR_REQUIRE(workspace, MACR_E_INVALID, "score_topk: workspace is null (macr_score_topk_workspace_bytes)");
    MACR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID,
                 "score_topk: workspace must be 256-byte aligned");
    const StreamGeo geo = stream_geo(U, n_local, d);
    TopkWs ws = carve_topk_ws(workspace, U, n_local, geo, d);
    MACR_REQUIRE(workspace_bytes >= ws.bytes, MACR_E_WORKSPACE, "score_topk: workspace %zu < %zu bytes", workspace_bytes,
                 ws.bytes);
    // workspace head: counts, flag, shared_thr <- 0; tau <- -inf on the list-everything path; maxima <- NaN
    const bool list_all = n_local <= kSelRegs * 64 && geo.range1 * kTileItems <= ws.cap;
    // (a shard small enough to list everything has nothing to filter)
    const bool filter_bf16 = !list_all && eval_filter_bf16(filter);
    const bool half = filter_bf16 && eval_filter_half(filter);          // the fp16 filter's kernels on the reduced-precision path
    // class maxima start as NaN (= no unmasked item seen).  Under the bf16 filter the sampling pass may keep 16 per split and
    // query instead of 32 (merge_pairs below), and a seeded first round has no sampling pass at all: its repair round, if one
    // follows (macr_score_topk_repair_round), fills them itself.
    const bool bf16_merge = filter_bf16 && geo.slots0 * 16 >= 4 * K;
    const size_t n_max_words = list_all ? 0 : filter_bf16 ? (size_t)geo.slots0 * U * (bf16_merge ? 16 : 32) : ws.maxima_bytes / 4;
    if (repair_only && filter_bf16 && seed_idx != nullptr) fill_words(reinterpret_cast<uint32_t *>(ws.maxima), n_max_words, 0xffffffffu, st);
    if (!repair_only && !ws_ready) {
        const WsHeadPlan hp = ws_head_plan(ws, geo, U, n_local, K, filter, first_only && seed_idx != nullptr);
        k_topk_ws_init<<<hp.grid, 256, 0, st>>>(static_cast<uint32_t *>(workspace), hp.n_zero, hp.n_tau, 0xff800000u, hp.set_tau, hp.n_max);
        MACR_CHECK_LAUNCH("ws_init", st);
    }
    const int sel_blocks = (U + kSelWaves - 1) / kSelWaves;
    const uint32_t *mask_bits = mask_bits_in;
    MACR_REQUIRE(!mask_bits_in || mask_ptr, MACR_E_INVALID, "score_topk: mask_bits without the CSR mask it was built from");
    if (mask_ptr && !mask_bits_in) {
        if (!repair_only) {
            fill_words(ws.mask_bits, ws.mask_bytes / 4, 0u, st);
            k_mask_bits<<<(U + 3) / 4, 256, 0, st>>>(U, n_local, mask_ptr, mask_idx, item_offset, ws.mask_bits, n_tiles(n_local),
                                                     sample_log2(n_local));
            MACR_CHECK_LAUNCH("mask_bits", st);
        }
        mask_bits = ws.mask_bits;
    }
    if (repair_only && (K > MACR_MAX_TOPK_FUSED || list_all)) {
        // (the wide ranking and a shard that lists everything have no second round: the first call's result is final)
        if (stats) fill_words(stats, 2, 0u, st);
        return MACR_OK;
    }
    if (K > MACR_MAX_TOPK_FUSED) {
        // The wide ranking (--Ks up to 128: macr_mf/parse.py:31, utility/parser.py:63 take any list): thresholds at rank ~8K
        // and 1024-candidate lists are built for K <= 32, so larger K take the reference's own route -- a block of dense
        // score rows (the fp32 MFMA chain and epilogue of every other kernel here: the same bits), masked columns are no
        // candidates, the best K per row by the streaming selection of macr_topk_scores -- block by block through the list
        // buffer.  Same result contract: (score, global id) pairs in out[0], the other splits empty; seeds are not used.
        const size_t row_bytes = (size_t)n_local * 4;
        const int rows = (int)std::min<size_t>((size_t)U, ws.lists_bytes / row_bytes);
        MACR_REQUIRE(rows >= 1, MACR_E_WORKSPACE, "score_topk: K=%d needs %zu B of list buffer per query, have %zu", K, row_bytes, ws.lists_bytes);
        float *scores = reinterpret_cast<float *>(ws.lists);
        for (int q0 = 0; q0 < U; q0 += rows) {
            const int nq = std::min(rows, U - q0);
            dim3 grid((n_local + 127) / 128, (nq + 31) / 32);
            MACR_DISPATCH_DK(d, score_kind, (k_score_matrix<D, KIND><<<grid, 256, 0, st>>>(
                                                nq, n_local, user_ids ? users_tab : users_tab + (size_t)q0 * d, user_ids ? user_ids + q0 : nullptr,
                                                items, sig_u ? sig_u + q0 : nullptr, sig_i, c, c_dev, scores)));
            MACR_CHECK_LAUNCH("score_matrix", st);
            k_topk_scores_wide<<<(nq + 3) / 4, 256, 0, st>>>(scores, n_local, nq, K, out_idx + (size_t)q0 * K, out_val + (size_t)q0 * K,
                                                             mask_bits, U, q0, item_offset, K, 0, nullptr, nullptr);
            MACR_CHECK_LAUNCH("topk_scores", st);
        }
        if (n_splits > 1) {
            fill_words(out_val + (size_t)U * K, (size_t)(n_splits - 1) * U * K, 0xff800000u, st);
            fill_words(out_idx + (size_t)U * K, (size_t)(n_splits - 1) * U * K, 0xffffffffu, st);
        }
        if (stats) fill_words(stats, 2, 0u, st);
        return MACR_OK;
    }
    MACR_DISPATCH_DK(d, score_kind, {
        auto pass0 = k_score_stream<D, KIND, kModeMax>;
        auto pass1 = k_score_stream<D, KIND, kModeList>;
        auto pass0r = k_score_stream<D, KIND, kModeMax, 1, true>;
        auto pass1r = k_score_stream<D, KIND, kModeList, 1, true>;
        const size_t smem = StreamCfg<D>::smem;
        hipError_t e = hipSuccess;
        for (const void *f : {reinterpret_cast<const void *>(pass0), reinterpret_cast<const void *>(pass1),
                              reinterpret_cast<const void *>(pass0r), reinterpret_cast<const void *>(pass1r)})
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        MACR_REQUIRE(e == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS: %s", smem, hipGetErrorString(e));
        const int tau_regs = (geo.slots0 * 32 + 63) / 64;
        auto launch_tau = [&](bool rep) {
            if (rep) launch_k_tau<true>(tau_regs, sel_blocks, st, U, geo.slots0, K, ws.maxima, ws.blk_flag, ws.tau);
            else launch_k_tau<false>(tau_regs, sel_blocks, st, U, geo.slots0, K, ws.maxima, nullptr, ws.tau);
        };
        const bool seeded = !list_all && seed_idx;
        static const bool f32_sample_env = getenv("MACR_EVAL_F32_SAMPLE") && (getenv("MACR_EVAL_F32_SAMPLE")[0] == 'f' || getenv("MACR_EVAL_F32_SAMPLE")[0] == 'F');
        const bool f32_sample_bf16 = !filter_bf16 && !f32_sample_env;      // fp32 filter: thresholds from the bf16 sampling pass (see below)
        uint32_t *qmax_bits = reinterpret_cast<uint32_t *>(ws.overflow + 8);
        // users_c: the scaled query copies exist for DIRECT_MINUS_BOTH only; the other kinds' are the plain ones
        // (the fp16 copies have a layout of their own: always in users_c)
        uint4 *users_c = (half || KIND == MACR_SCORE_DIRECT_MINUS_BOTH) ? ws.users_c : ws.users_bf;
        const int n_part = (int)bf16_prep_c_blocks(U, n_local, D, kPrepTripsAlone);         // k_eval_prologue_prep's per-block item norms
        const uint32_t *zero_word = reinterpret_cast<const uint32_t *>(ws.overflow + 3);
        // k_score_sample_c merges its classes in pairs where that still leaves several times K of them (k_tau ranks half as many)
        const int merge_pairs = geo.slots0 * 16 >= 4 * K ? 1 : 0, per_c = merge_pairs ? 16 : 32;
        // Round 1 flags overflowing lists per user; the repair round lists the user blocks of those users again with the
        // threshold their cut lists imply (k_repair_plan); only a second overflow arms the exact fallback kernel.
        const bool repair = !list_all;
        if (!repair_only) {                                   // ---- the first round
        if (filter_bf16 && seeded) {
            // the listing pass's operand copies (the epilogue in the operands), |u| per query, max |q| -- and, in the same
            // launch, the seeded thresholds
            const unsigned n_prep = bf16_prep_c_blocks(U, n_local, D, kPrepTrips);
            if (prep_ready) {
                // (the copies exist: the seeded thresholds alone, plus the block that reduces the prologue's item norms to max |q|)
                k_tau_seed<D, KIND><<<(U + 7) / 8 + 1, 256, 0, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev,
                                                                       mask_bits, item_offset, K, seed_idx, ws.tau, ws.qpart, n_part, qmax_bits,
                                                                       mask_ptr, mask_idx);
                MACR_CHECK_LAUNCH("tau_seed", st);
            } else {
            auto prep_seed = half ? k_prep_tau_seed<D, KIND, true> : k_prep_tau_seed<D, KIND, false>;
            prep_seed<<<n_prep + (U + 7) / 8, 256, 0, st>>>((int)n_prep, U, n_local, users_tab, user_ids, items, users_c,
                                                            ws.items_c, ws.unorm, qmax_bits, sig_u, sig_i, c, c_dev, mask_bits,
                                                            item_offset, K, seed_idx, ws.tau, mask_ptr, mask_idx);
            MACR_CHECK_LAUNCH("bf16_prep+tau_seed", st);
            }
        } else if (filter_bf16 && prep_ready) {
            // (the copies exist; the sampling pass below reduces the item norms)
        } else if (filter_bf16) {
            auto prep = half ? k_bf16_prep_c<D, KIND, true> : k_bf16_prep_c<D, KIND, false>;
            prep<<<bf16_prep_c_blocks(U, n_local, D, kPrepTripsAlone), 256, 0, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev,
                                                                                   users_c, ws.items_c, ws.unorm, qmax_bits);
            MACR_CHECK_LAUNCH("bf16_prep_c", st);
        }
        if (filter_bf16 && !seeded) {
            auto pass0c = half ? k_score_sample_c<D, KIND, false, true> : k_score_sample_c<D, KIND, false, false>;
            const size_t smem_c = half ? StreamCfgH1<D>::smem : StreamCfgC<D>::smem;
            MACR_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void *>(pass0c), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)smem_c) == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS", smem_c);
            pass0c<<<geo.grid0, 512, smem_c, st>>>(U, n_local, users_c, ws.items_c, sig_u, mask_bits, zero_word,
                                                  geo.ublocks, ws.maxima, sample_log2(n_local), merge_pairs, nullptr, nullptr,
                                                  prep_ready ? ws.qpart : nullptr, n_part, qmax_bits);
            MACR_CHECK_LAUNCH("score_sample_b", st);
            launch_k_tau<false>((geo.slots0 * per_c + 63) / 64, sel_blocks, st, U, geo.slots0, K, ws.maxima, nullptr, ws.tau, ws.unorm, qmax_bits, c, c_dev, D,
                                merge_pairs ? 4 : 5, half ? 1 : 0, (half && KIND == MACR_SCORE_RUBI_BOTH) ? sig_u : nullptr);
            MACR_CHECK_LAUNCH("tau", st);
        } else if (seeded && filter_bf16) {
            // (thresholds: in the launch above)
        } else if (seeded) {
            // thresholds from the exact scores of the caller's seed items (its previous top K): no sampling pass, no k_tau
            k_tau_seed<D, KIND><<<(U + 7) / 8, 256, 0, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev,
                                                               mask_bits, item_offset, K, seed_idx, ws.tau, nullptr, 0, nullptr, mask_ptr, mask_idx);
            MACR_CHECK_LAUNCH("tau_seed", st);
        } else if (!list_all && f32_sample_bf16) {
            // fp32 listing, thresholds from the reduced-precision sampling pass (round 6).  k_tau's threshold under that pass is already
            // a statement about fp32 scores -- "K sampled items score >= tau in fp32" (the filter's margin is taken off there) --
            // which is all the fp32 listing pass asks of it.  The pass runs on the FP16 copies (one MFMA per 16 k): operand copies +
            // sampling pass + k_tau cost 9 + 28 + 13 us at the Gowalla shape where the bf16 copies cost 12 + 46 + 11 and the fp32
            // sampling pass + k_tau 104 + 20.  The ranking is the fp32 ranking either way (tests/ under every filter);
            // MACR_EVAL_F32_SAMPLE=f32 keeps the fp32 sampling pass (A/B).  The class maxima were NaN-filled for the 32-per-class
            // layout, which covers the 16-per-class one the merged sampling pass writes.
            k_bf16_prep_c<D, KIND, true><<<bf16_prep_c_blocks(U, n_local, D, kPrepTripsAlone), 256, 0, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev,
                                                                                            ws.users_c, ws.items_c, ws.unorm, qmax_bits);
            MACR_CHECK_LAUNCH("bf16_prep_c", st);
            auto pass0c = k_score_sample_c<D, KIND, false, true>;
            const size_t smem_c = StreamCfgH1<D>::smem;
            MACR_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void *>(pass0c), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)smem_c) == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS", smem_c);
            pass0c<<<geo.grid0, 512, smem_c, st>>>(U, n_local, ws.users_c, ws.items_c, sig_u, mask_bits, zero_word,
                                                  geo.ublocks, ws.maxima, sample_log2(n_local), merge_pairs, nullptr, nullptr, nullptr, 0, nullptr);
            MACR_CHECK_LAUNCH("score_sample_b", st);
            launch_k_tau<false>((geo.slots0 * per_c + 63) / 64, sel_blocks, st, U, geo.slots0, K, ws.maxima, nullptr, ws.tau, ws.unorm, qmax_bits, c, c_dev, D,
                                merge_pairs ? 4 : 5, 1, KIND == MACR_SCORE_RUBI_BOTH ? sig_u : nullptr);
            MACR_CHECK_LAUNCH("tau", st);
        } else if (!list_all) {
            // (a catalogue (shard) whose every tile range fits a candidate list needs no threshold: tau = -inf lists every
            // unmasked item and the selection kernel ranks them -- no sampling pass, no k_tau)
            pass0<<<geo.grid0, 512, smem, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_bits, item_offset,
                                                geo.ublocks, ws.tau, ws.maxima, ws.lists, ws.counts, ws.cap, ws.overflow, 0,
                                                sample_log2(n_local), nullptr, nullptr, nullptr, SweepArgs{}, 0);
            MACR_CHECK_LAUNCH("score_sample", st);
            launch_tau(false);
            MACR_CHECK_LAUNCH("tau", st);
        }
        if (filter_bf16) {
            // bf16-filtered first round (k_score_stream_b): listing on the bf16 matrix cores, then the selection re-scores
            // its best candidates in fp32
            const void *pass1c = half ? reinterpret_cast<const void *>(k_score_stream_h<D, KIND>) : reinterpret_cast<const void *>(k_score_stream_c<D, KIND>);
            const size_t smem_c = half ? StreamCfgH<D>::smem : StreamCfgC<D>::smem;
            MACR_REQUIRE(hipFuncSetAttribute(pass1c, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)smem_c) == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS", smem_c);
            if (half)
                k_score_stream_h<D, KIND><<<geo.grid_h, StreamCfgH<D>::THREADS, smem_c, st>>>(U, n_local, users_c, ws.items_c, ws.unorm, qmax_bits, sig_u, c, c_dev,
                                                                       mask_bits, zero_word, item_offset, geo.ublocks,
                                                                       ws.tau, ws.lists, ws.counts, ws.cap,
                                                                       repair ? ws.user_ovf : ws.overflow, repair ? 1 : 0,
                                                                       seeded ? ws.blk_flag : nullptr, nullptr, nullptr, 0);
            else
                k_score_stream_c<D, KIND><<<geo.grid1, 512, smem_c, st>>>(U, n_local, users_c, ws.items_c, ws.unorm, qmax_bits, sig_u, c, c_dev,
                                                                       mask_bits, zero_word, item_offset, geo.ublocks,
                                                                       ws.tau, ws.lists, ws.counts, ws.cap,
                                                                       repair ? ws.user_ovf : ws.overflow, repair ? 1 : 0,
                                                                       seeded ? ws.blk_flag : nullptr, nullptr, nullptr, 0);
            MACR_CHECK_LAUNCH("score_stream_b", st);
            k_select_b<D, KIND><<<sel_blocks, 64 * kSelWaves, 0, st>>>(U, n_local, geo.slots1, n_splits, K, ws.cap, ws.lists, ws.counts,
                                                                      repair ? ws.user_ovf : ws.overflow, repair ? 1 : 0, nullptr,
                                                                      seeded ? ws.blk_flag : nullptr, users_tab, user_ids, items,
                                                                      sig_u, sig_i, c, c_dev, item_offset, ws.unorm, qmax_bits,
                                                                      out_val, out_idx, seed_out, half ? 1 : 0);
            MACR_CHECK_LAUNCH("select_b", st);
        } else {
        pass1<<<geo.grid1, 512, smem, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_bits, item_offset,
                                            geo.ublocks, ws.tau, ws.maxima, ws.lists, ws.counts, ws.cap,
                                            repair ? ws.user_ovf : ws.overflow, repair ? 1 : 0, 0, nullptr, nullptr,
                                            seeded ? ws.blk_flag : nullptr, SweepArgs{}, 0);
        MACR_CHECK_LAUNCH("score_stream", st);
        k_select<false><<<sel_blocks, 64 * kSelWaves, 0, st>>>(U, geo.slots1, n_splits, K, ws.cap, ws.lists, ws.counts,
                                                              repair ? ws.user_ovf : ws.overflow, repair ? 1 : 0, nullptr,
                                                              seeded ? ws.blk_flag : nullptr, out_val, out_idx, seed_out);
        MACR_CHECK_LAUNCH("select", st);
        }
        if (repair) {
            k_repair_plan<<<geo.ublocks, kUsersPerBlock, 0, st>>>(U, K, geo.slots1, ws.user_ovf, out_val, ws.tau, ws.counts,
                                                                 ws.ub_map, ws.blk_flag, ws.overflow + 1, ws.overflow + 2,
                                                                 first_only ? stats : nullptr);
            MACR_CHECK_LAUNCH("repair_plan", st);
        }
        }                                                     // ---- (first round)
        if (repair) {
            if (first_only) {
                // (stats[0] = query blocks whose lists overflowed or whose seeds were stale: their rows of out_* are not the ranking)
            } else if (filter_bf16) {
                // the repair round on the bf16 copies too: sampling pass for the re-listed user blocks when the thresholds
                // came from seeds, listing, selection with fp32 re-scoring
                auto pass0rb = half ? k_score_sample_c<D, KIND, true, true> : k_score_sample_c<D, KIND, true, false>;
                const void *pass1rc = half ? reinterpret_cast<const void *>(k_score_stream_h<D, KIND, true>) : reinterpret_cast<const void *>(k_score_stream_c<D, KIND, true>);
                const size_t smem_c0 = half ? StreamCfgH1<D>::smem : StreamCfgC<D>::smem, smem_c = half ? StreamCfgH<D>::smem : StreamCfgC<D>::smem;
                hipError_t eb = hipFuncSetAttribute(reinterpret_cast<const void *>(pass0rb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_c0);
                if (eb == hipSuccess) eb = hipFuncSetAttribute(pass1rc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_c);
                MACR_REQUIRE(eb == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS", smem_c);
                if (seeded) {
                    pass0rb<<<geo.grid0, 512, smem_c0, st>>>(U, n_local, users_c, ws.items_c, sig_u, mask_bits, zero_word,
                                                            geo.ublocks, ws.maxima, sample_log2(n_local), merge_pairs, ws.ub_map, ws.overflow + 1,
                                                            nullptr, 0, nullptr);
                    MACR_CHECK_LAUNCH("score_sample2", st);
                    launch_k_tau<true>((geo.slots0 * per_c + 63) / 64, sel_blocks, st, U, geo.slots0, K, ws.maxima, ws.blk_flag, ws.tau, ws.unorm, qmax_bits, c, c_dev, D,
                                       merge_pairs ? 4 : 5, half ? 1 : 0, (half && KIND == MACR_SCORE_RUBI_BOTH) ? sig_u : nullptr);
                    MACR_CHECK_LAUNCH("tau2", st);
                }
                if (half)
                    k_score_stream_h<D, KIND, true><<<geo.grid_h, StreamCfgH<D>::THREADS, smem_c, st>>>(U, n_local, users_c, ws.items_c, ws.unorm, qmax_bits, sig_u, c, c_dev,
                                                                                mask_bits, zero_word, item_offset, geo.ublocks, ws.tau,
                                                                                ws.lists, ws.counts, ws.cap, ws.overflow, 0, nullptr,
                                                                                ws.ub_map, ws.overflow + 1, geo.slots1);
                else
                    k_score_stream_c<D, KIND, true><<<geo.grid1, 512, smem_c, st>>>(U, n_local, users_c, ws.items_c, ws.unorm, qmax_bits, sig_u, c, c_dev,
                                                                                mask_bits, zero_word, item_offset, geo.ublocks, ws.tau,
                                                                                ws.lists, ws.counts, ws.cap, ws.overflow, 0, nullptr,
                                                                                ws.ub_map, ws.overflow + 1, geo.slots1);
                MACR_CHECK_LAUNCH("score_stream2", st);
                k_select_b<D, KIND, true><<<sel_blocks, 64 * kSelWaves, 0, st>>>(U, n_local, geo.slots1, n_splits, K, ws.cap, ws.lists,
                                                                                ws.counts, ws.overflow, 0, ws.overflow + 1, ws.blk_flag,
                                                                                users_tab, user_ids, items, sig_u, sig_i, c, c_dev,
                                                                                item_offset, ws.unorm, qmax_bits, out_val, out_idx, seed_out, half ? 1 : 0);
                MACR_CHECK_LAUNCH("select2", st);
            } else {
            if (seeded) {
                pass0r<<<geo.grid0, 512, smem, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_bits,
                                                     item_offset, geo.ublocks, ws.tau, ws.maxima, ws.lists, ws.counts, ws.cap,
                                                     ws.overflow, 0, sample_log2(n_local), ws.ub_map, ws.overflow + 1, nullptr, SweepArgs{}, 0);
                MACR_CHECK_LAUNCH("score_sample2", st);
                launch_tau(true);
                MACR_CHECK_LAUNCH("tau2", st);
            }
            pass1r<<<geo.grid1, 512, smem, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_bits, item_offset,
                                                 geo.ublocks, ws.tau, ws.maxima, ws.lists, ws.counts, ws.cap, ws.overflow, 0, 0,
                                                 ws.ub_map, ws.overflow + 1, nullptr, SweepArgs{}, geo.slots1);
            MACR_CHECK_LAUNCH("score_stream2", st);
            k_select<true><<<sel_blocks, 64 * kSelWaves, 0, st>>>(U, geo.slots1, n_splits, K, ws.cap, ws.lists, ws.counts, ws.overflow, 0,
                                                           ws.overflow + 1, ws.blk_flag, out_val, out_idx, seed_out);
            MACR_CHECK_LAUNCH("select2", st);
            }
        }
    });
    if (first_only && !list_all) return MACR_OK;
    if (first_only) {
        // a shard small enough to list everything cannot overflow a list: the first round is the ranking
        fill_words(stats, 2, 0u, st);
        return MACR_OK;
    }
    // Fallback, armed by the overflow flag on the device (its blocks return at once otherwise): the running
    // top-K kernel is exact for any score order.
    const size_t smem_old = score_topk_smem_bytes();
    MACR_DISPATCH_DK(d, score_kind, {
        auto kern = k_score_topk<D, KIND>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_old);
        MACR_REQUIRE(e == hipSuccess, MACR_E_LAUNCH, "score_topk: cannot reserve %zu B of LDS: %s", smem_old, hipGetErrorString(e));
        kern<<<ublocks * n_splits, 512, smem_old, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev,
                                                        mask_ptr, mask_idx, item_offset, K, n_splits, out_val, out_idx,
                                                        n_splits > 1 ? ws.shared_thr : nullptr, force_fallback ? nullptr : ws.overflow,
                                                        stats, force_fallback ? nullptr : ws.blk_flag);
    });
    MACR_CHECK_LAUNCH("score_topk", st);
    return MACR_OK;
}

extern "C" int macr_score_topk(int score_kind, int filter, int U, int n_local, int d, const float *users_tab,
                               const int32_t *user_ids, const float *items, const float *sig_u,
                               const float *sig_i, float c, const float *c_dev, const int32_t *mask_ptr, const int32_t *mask_idx,
                               const uint32_t *mask_bits_in, int item_offset, int K, int n_splits, const int32_t *seed_idx,
                               int32_t *seed_out, float *out_val, int32_t *out_idx, int32_t *stats, void *workspace,
                               size_t workspace_bytes, void *stream) {
    return score_topk_impl(0, filter, score_kind, U, n_local, d, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_ptr, mask_idx,
                           mask_bits_in, item_offset, K, n_splits, seed_idx, seed_out, out_val, out_idx, stats, workspace,
                           workspace_bytes, stream);
}

extern "C" int macr_score_topk_first_round(int score_kind, int filter, int U, int n_local, int d, const float *users_tab,
                                           const int32_t *user_ids, const float *items, const float *sig_u,
                                           const float *sig_i, float c, const float *c_dev, const int32_t *mask_ptr,
                                           const int32_t *mask_idx, const uint32_t *mask_bits_in, int item_offset, int K,
                                           int n_splits, const int32_t *seed_idx, int32_t *seed_out, float *out_val,
                                           int32_t *out_idx, int32_t *stats, void *workspace, size_t workspace_bytes,
                                           void *stream) {
    MACR_REQUIRE(stats, MACR_E_INVALID, "score_topk_first_round: stats is null (it says whether the result stands)");
    return score_topk_impl(1, filter, score_kind, U, n_local, d, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_ptr, mask_idx,
                           mask_bits_in, item_offset, K, n_splits, seed_idx, seed_out, out_val, out_idx, stats, workspace,
                           workspace_bytes, stream);
}

extern "C" int macr_score_topk_repair_round(int score_kind, int filter, int U, int n_local, int d, const float *users_tab,
                                            const int32_t *user_ids, const float *items, const float *sig_u,
                                            const float *sig_i, float c, const float *c_dev, const int32_t *mask_ptr,
                                            const int32_t *mask_idx, const uint32_t *mask_bits_in, int item_offset, int K,
                                            int n_splits, const int32_t *seed_idx, int32_t *seed_out, float *out_val,
                                            int32_t *out_idx, int32_t *stats, void *workspace, size_t workspace_bytes,
                                            void *stream) {
    return score_topk_impl(2, filter, score_kind, U, n_local, d, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, mask_ptr, mask_idx,
                           mask_bits_in, item_offset, K, n_splits, seed_idx, seed_out, out_val, out_idx, stats, workspace,
                           workspace_bytes, stream);
}

extern "C" int macr_score_topk_prologue(int filter, int U, int n_local, int d, int K, int seeded_first_round,
                                        const float *items, const float *w_item, float *sig_i,
                                        const float *users_tab, const int32_t *user_ids, const float *w_user, float *sig_u,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(filter >= MACR_EVAL_FILTER_ENV && filter <= MACR_EVAL_FILTER_F16, MACR_E_INVALID, "score_topk_prologue: filter=%d", filter);
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_topk_prologue: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_topk_prologue: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "score_topk_prologue: K=%d outside [1,%d]", K, MACR_MAX_TOPK);
    MACR_REQUIRE(items && w_item && sig_i, MACR_E_INVALID, "score_topk_prologue: null pointer (items, w_item, sig_i)");
    MACR_REQUIRE((sig_u == nullptr) == (w_user == nullptr) && (!sig_u || users_tab), MACR_E_INVALID,
                 "score_topk_prologue: sig_u, w_user and users_tab come together");
    MACR_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID,
                 "score_topk_prologue: workspace is null or not 256-byte aligned");
    const StreamGeo geo = stream_geo(U, n_local, d);
    TopkWs ws = carve_topk_ws(workspace, U, n_local, geo, d);
    MACR_REQUIRE(workspace_bytes >= ws.bytes, MACR_E_WORKSPACE, "score_topk_prologue: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    const WsHeadPlan hp = ws_head_plan(ws, geo, U, n_local, K, filter, seeded_first_round != 0);
    MACR_DISPATCH_LPR(d, {
        const int rpb = 256 / LPR, nb_a = (n_local + rpb - 1) / rpb, nb_b = sig_u ? (U + rpb - 1) / rpb : 0;
        k_eval_prologue<LPR><<<nb_a + nb_b + hp.grid, 256, 0, st>>>(nb_a, items, n_local, w_item, sig_i, nb_b, users_tab, user_ids, U, w_user, sig_u,
                                                                   static_cast<uint32_t *>(workspace), hp.n_zero, hp.n_tau, 0xff800000u,
                                                                   hp.set_tau, hp.n_max);
    });
    MACR_CHECK_LAUNCH("eval_prologue", st);
    return MACR_OK;
}

// The same plus the fp16 filter's operand copies (abi 15): one launch reads every row once.  The ranking call that follows takes
// MACR_EVAL_WS_READY | MACR_EVAL_PREP_READY with MACR_EVAL_FILTER_F16 and the same score_kind, c, tables.
extern "C" int macr_score_topk_prologue_prep(int score_kind, int U, int n_local, int d, int K, int seeded_first_round,
                                             const float *items, const float *w_item, float *sig_i,
                                             const float *users_tab, const int32_t *user_ids, const float *w_user, float *sig_u,
                                             float c, const float *c_dev, void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(score_kind_valid(score_kind), MACR_E_INVALID, "score_topk_prologue_prep: score_kind=%d", score_kind);
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_topk_prologue_prep: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_topk_prologue_prep: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK_FUSED, MACR_E_UNSUPPORTED, "score_topk_prologue_prep: K=%d outside [1,%d] (the fused ranking)", K,
                 MACR_MAX_TOPK_FUSED);
    MACR_REQUIRE(items && w_item && sig_i && users_tab, MACR_E_INVALID, "score_topk_prologue_prep: null pointer (items, w_item, sig_i, users_tab)");
    MACR_REQUIRE((sig_u == nullptr) == (w_user == nullptr), MACR_E_INVALID, "score_topk_prologue_prep: sig_u and w_user come together");
    MACR_REQUIRE(!score_uses_sig_u(score_kind) || sig_u, MACR_E_INVALID, "score_topk_prologue_prep: score_kind %d needs sig_u / w_user", score_kind);
    MACR_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID,
                 "score_topk_prologue_prep: workspace is null or not 256-byte aligned");
    const StreamGeo geo = stream_geo(U, n_local, d);
    TopkWs ws = carve_topk_ws(workspace, U, n_local, geo, d);
    MACR_REQUIRE(workspace_bytes >= ws.bytes, MACR_E_WORKSPACE, "score_topk_prologue_prep: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    const bool list_all = n_local <= kSelRegs * 64 && geo.range1 * kTileItems <= ws.cap;
    MACR_REQUIRE(!list_all, MACR_E_UNSUPPORTED, "score_topk_prologue_prep: a catalogue this small lists everything -- no filter, no copies (use macr_score_topk_prologue)");
    const WsHeadPlan hp = ws_head_plan(ws, geo, U, n_local, K, MACR_EVAL_FILTER_F16, seeded_first_round != 0);
    MACR_DISPATCH_DK(d, score_kind, {
        const unsigned n_prep = bf16_prep_c_blocks(U, n_local, D, kPrepTripsAlone);
        k_eval_prologue_prep<D, KIND><<<n_prep + hp.grid, 256, 0, st>>>((int)n_prep, U, n_local, users_tab, user_ids, items, w_item, w_user, sig_i, sig_u,
                                                                       c, c_dev, ws.users_c, ws.items_c, ws.unorm, ws.qpart,
                                                                       static_cast<uint32_t *>(workspace), hp.n_zero, hp.n_tau, 0xff800000u,
                                                                       hp.set_tau, hp.n_max);
    });
    MACR_CHECK_LAUNCH("eval_prologue_prep", st);
    return MACR_OK;
}

/* ---- c sweep -------------------------------------------------------------------------------------------------- */
extern "C" size_t macr_score_topk_sweep_workspace_bytes(int U, int n_local, int d, int n_c) {
    if (U <= 0 || n_local <= 0 || !dim_supported(d) || n_c < 1 || n_c > kMaxSweep) return 0;
    const StreamGeo geo = stream_geo(U, n_local, d);
    return (size_t)n_c * align_up(carve_topk_ws(nullptr, U, n_local, geo, d).bytes, 256);
}

extern "C" int macr_score_topk_sweep(int score_kind, int filter, int U, int n_local, int d, const float *users_tab,
                                     const int32_t *user_ids, const float *items, const float *sig_u, const float *sig_i,
                                     int n_c, const float *c_dev, const int32_t *mask_ptr, const int32_t *mask_idx,
                                     const uint32_t *mask_bits_in, int item_offset, int K, float *out_val, int32_t *out_idx,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    MACR_REQUIRE(score_kind_valid(score_kind) && score_uses_sig_i(score_kind), MACR_E_INVALID,
                 "score_topk_sweep: score_kind=%d does not depend on c", score_kind);
    MACR_REQUIRE(filter >= MACR_EVAL_FILTER_ENV && filter <= MACR_EVAL_FILTER_F16, MACR_E_INVALID, "score_topk_sweep: filter=%d", filter);
    MACR_REQUIRE(n_c >= 1 && n_c <= kMaxSweep, MACR_E_UNSUPPORTED, "score_topk_sweep: n_c=%d outside [1,%d]", n_c, kMaxSweep);
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_topk_sweep: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_topk_sweep: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(K >= 1 && K <= MACR_MAX_TOPK_FUSED, MACR_E_UNSUPPORTED, "score_topk_sweep: K=%d outside [1,%d] (larger K: macr_score_topk per value)", K, MACR_MAX_TOPK_FUSED);
    MACR_REQUIRE(users_tab && items && out_val && out_idx && c_dev && sig_i && (!score_uses_sig_u(score_kind) || sig_u),
                 MACR_E_INVALID, "score_topk_sweep: null pointer");
    MACR_REQUIRE(mask_bits_in || !mask_ptr, MACR_E_INVALID, "score_topk_sweep: pass the mask bitmap (macr_mask_bits_build) with the mask");
    MACR_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, MACR_E_INVALID, "score_topk_sweep: workspace");
    const StreamGeo geo = stream_geo(U, n_local, d);
    const size_t one = align_up(carve_topk_ws(nullptr, U, n_local, geo, d).bytes, 256);
    MACR_REQUIRE(workspace_bytes >= one * n_c, MACR_E_WORKSPACE, "score_topk_sweep: workspace %zu < %zu bytes", workspace_bytes, one * n_c);
    static const bool force_fallback = getenv("MACR_TOPK_FALLBACK") && getenv("MACR_TOPK_FALLBACK")[0] == '1';
    hipStream_t st = as_stream(stream);
    const int ublocks = (U + kUsersPerBlock - 1) / kUsersPerBlock, sel_blocks = (U + kSelWaves - 1) / kSelWaves;
    const bool list_all = n_local <= kSelRegs * 64 && geo.range1 * kTileItems <= carve_topk_ws(nullptr, U, n_local, geo).cap;
    TopkWs ws[kMaxSweep];
    SweepArgs sw;
    sw.n_c = n_c;
    for (int g = 0; g < kMaxSweep; ++g) {
        ws[g] = carve_topk_ws(static_cast<char *>(workspace) + (size_t)(g < n_c ? g : 0) * one, U, n_local, geo, d);
        sw.tau[g] = ws[g].tau; sw.lists[g] = ws[g].lists; sw.counts[g] = ws[g].counts; sw.overflow[g] = ws[g].overflow;
    }
    MACR_DISPATCH_DK(d, score_kind, {
        auto pass0 = k_score_stream<D, KIND, kModeMax>;
        auto pass1 = k_score_stream<D, KIND, kModeList, kMaxSweep>;
        const size_t smem = StreamCfg<D>::smem, smem1 = StreamCfg<D>::smem_sweep;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pass0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(pass1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
        MACR_REQUIRE(e == hipSuccess, MACR_E_LAUNCH, "score_topk_sweep: cannot reserve %zu B of LDS: %s", smem1, hipGetErrorString(e));
        const bool filter_bf16 = !list_all && eval_filter_bf16(filter);
        uint32_t *qmax_bits = reinterpret_cast<uint32_t *>(ws[0].overflow + 8);      // (value 0's workspace holds the operand copies)
        const size_t smem_b = StreamCfgB<D>::smem, smem_bs = StreamCfgB<D>::smem + (size_t)(kMaxSweep - 1) * kUsersPerBlock * 4;
        if (filter_bf16) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_sample_b<D, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_stream_bs<D, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bs);
            MACR_REQUIRE(e == hipSuccess, MACR_E_LAUNCH, "score_topk_sweep: cannot reserve %zu B of LDS: %s", smem_bs, hipGetErrorString(e));
        }
        for (int g = 0; g < n_c; ++g) {           // per c: workspace head, sampling pass, threshold
            const size_t n_zero = ws[g].header_bytes / 4;
            const size_t n_tau = (reinterpret_cast<char *>(ws[g].maxima) - reinterpret_cast<char *>(ws[g].tau)) / 4;
            const size_t n_max = list_all ? 0 : ws[g].maxima_bytes / 4;
            const size_t total = n_zero + n_tau + n_max;
            const unsigned grid = (unsigned)((total + 256 * 8 - 1) / (256 * 8) < 2048 ? (total + 256 * 8 - 1) / (256 * 8) : 2048);
            k_topk_ws_init<<<grid ? grid : 1, 256, 0, st>>>(reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + (size_t)g * one),
                                                         n_zero, n_tau, 0xff800000u, list_all ? 1 : 0, n_max);
            MACR_CHECK_LAUNCH("ws_init", st);
            if (list_all) continue;
            if (filter_bf16) {
                if (g == 0) {
                    k_bf16_prep<D><<<bf16_prep_blocks(U, n_local, D), 256, 0, st>>>(
                        U, n_local, users_tab, user_ids, items, ws[0].users_bf, ws[0].items_bf, ws[0].unorm, qmax_bits);
                    MACR_CHECK_LAUNCH("bf16_prep", st);
                }
                k_score_sample_b<D, KIND><<<geo.grid0, 512, smem_b, st>>>(U, n_local, ws[0].users_bf, ws[0].items_bf, sig_u, sig_i, 0.f, c_dev + g,
                                                                         mask_bits_in, geo.ublocks, ws[g].maxima, sample_log2(n_local), nullptr, nullptr);
                MACR_CHECK_LAUNCH("score_sample_b", st);
                launch_k_tau<false>((geo.slots0 * 32 + 63) / 64, sel_blocks, st, U, geo.slots0, K, ws[g].maxima, nullptr, ws[g].tau, ws[0].unorm,
                                    qmax_bits, 0.f, c_dev + g, D);
                MACR_CHECK_LAUNCH("tau", st);
                continue;
            }
            pass0<<<geo.grid0, 512, smem, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, 0.f, c_dev + g, mask_bits_in,
                                                item_offset, geo.ublocks, ws[g].tau, ws[g].maxima, ws[g].lists, ws[g].counts,
                                                ws[g].cap, ws[g].overflow, 0, sample_log2(n_local), nullptr, nullptr, nullptr, SweepArgs{}, 0);
            MACR_CHECK_LAUNCH("score_sample", st);
            const int tau_regs = (geo.slots0 * 32 + 63) / 64;
            launch_k_tau<false>(tau_regs, sel_blocks, st, U, geo.slots0, K, ws[g].maxima, nullptr, ws[g].tau);
            MACR_CHECK_LAUNCH("tau", st);
        }
        // ONE listing pass for all n_c values
        if (filter_bf16) {
            k_score_stream_bs<D, KIND><<<geo.grid1, 512, smem_bs, st>>>(U, n_local, ws[0].users_bf, ws[0].items_bf, ws[0].unorm, qmax_bits, sig_u,
                                                                       sig_i, c_dev, mask_bits_in, item_offset, geo.ublocks, ws[0].cap, sw);
            MACR_CHECK_LAUNCH("score_stream_b", st);
            for (int g = 0; g < n_c; ++g) {
                k_select_b<D, KIND><<<sel_blocks, 64 * kSelWaves, 0, st>>>(U, n_local, geo.slots1, 1, K, ws[g].cap, ws[g].lists, ws[g].counts,
                                                                          ws[g].overflow, 0, nullptr, nullptr, users_tab, user_ids, items, sig_u,
                                                                          sig_i, 0.f, c_dev + g, item_offset, ws[0].unorm, qmax_bits,
                                                                          out_val + (size_t)g * U * K, out_idx + (size_t)g * U * K, nullptr, 0);
                MACR_CHECK_LAUNCH("select_b", st);
            }
        } else {
        pass1<<<geo.grid1, 512, smem1, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, 0.f, c_dev, mask_bits_in,
                                             item_offset, geo.ublocks, ws[0].tau, ws[0].maxima, ws[0].lists, ws[0].counts,
                                             ws[0].cap, ws[0].overflow, 0, sample_log2(n_local), nullptr, nullptr, nullptr, sw, 0);
        MACR_CHECK_LAUNCH("score_stream", st);
        for (int g = 0; g < n_c; ++g) {
            k_select<false><<<sel_blocks, 64 * kSelWaves, 0, st>>>(U, geo.slots1, 1, K, ws[g].cap, ws[g].lists, ws[g].counts, ws[g].overflow,
                                                           0, nullptr, nullptr, out_val + (size_t)g * U * K, out_idx + (size_t)g * U * K, nullptr);
            MACR_CHECK_LAUNCH("select", st);
        }
        }
    });
    // per-c fallback, armed by that c's overflow flag
    const size_t smem_old = score_topk_smem_bytes();
    MACR_DISPATCH_DK(d, score_kind, {
        auto kern = k_score_topk<D, KIND>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_old);
        MACR_REQUIRE(e == hipSuccess, MACR_E_LAUNCH, "score_topk_sweep: cannot reserve %zu B of LDS: %s", smem_old, hipGetErrorString(e));
        for (int g = 0; g < n_c; ++g)
            kern<<<ublocks, 512, smem_old, st>>>(U, n_local, users_tab, user_ids, items, sig_u, sig_i, 0.f, c_dev + g, mask_ptr, mask_idx,
                                                 item_offset, K, 1, out_val + (size_t)g * U * K, out_idx + (size_t)g * U * K,
                                                 nullptr, force_fallback ? nullptr : ws[g].overflow, nullptr, nullptr);
    });
    MACR_CHECK_LAUNCH("score_topk", st);
    return MACR_OK;
}

extern "C" int macr_score_matrix(int score_kind, int U, int n_local, int d, const float *users_tab,
                                 const int32_t *user_ids, const float *items, const float *sig_u,
                                 const float *sig_i, float c, const float *c_dev, float *out_scores, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(score_kind_valid(score_kind), MACR_E_INVALID, "score_matrix: score_kind=%d", score_kind);
    MACR_REQUIRE(U > 0 && n_local > 0, MACR_E_INVALID, "score_matrix: U=%d n_local=%d", U, n_local);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "score_matrix: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(users_tab && items && out_scores, MACR_E_INVALID, "score_matrix: null pointer");
    MACR_REQUIRE((!score_uses_sig_i(score_kind) || sig_i) && (!score_uses_sig_u(score_kind) || sig_u), MACR_E_INVALID,
                 "score_matrix: score_kind %d needs sig_i%s", score_kind, score_uses_sig_u(score_kind) ? " and sig_u" : "");
    dim3 grid((n_local + 127) / 128, (U + 31) / 32);
    MACR_DISPATCH_DK(d, score_kind, (k_score_matrix<D, KIND><<<grid, 256, 0, st>>>(
                                        U, n_local, users_tab, user_ids, items, sig_u, sig_i, c, c_dev, out_scores)));
    MACR_CHECK_LAUNCH("score_matrix", st);
    return MACR_OK;
}

#ifdef MACR_TEST_ENTRY_POINTS       // libmacr_hip_test.so only (include/macr_hip_test.h)
#include "../../include/macr_hip_test.h"
extern "C" size_t macr_test_bf16_products_workspace_bytes(int d, int U, int N) {
    if (!dim_supported(d) || U <= 0 || N <= 0) return 0;
    return align_up((size_t)U * 4 * d, 256) + align_up((size_t)N * 4 * d, 256) + align_up((size_t)U * 4, 256) + 256;
}

extern "C" int macr_test_bf16_products(int d, int U, int N, const float *users, const float *items, float c, float *prod,
                                       float *margin, void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && N > 0, MACR_E_INVALID, "test_bf16_products: U=%d N=%d", U, N);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "test_bf16_products: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(users && items && prod && margin && workspace, MACR_E_INVALID, "test_bf16_products: null pointer");
    MACR_REQUIRE(workspace_bytes >= macr_test_bf16_products_workspace_bytes(d, U, N), MACR_E_WORKSPACE,
                 "test_bf16_products: workspace %zu < %zu", workspace_bytes, macr_test_bf16_products_workspace_bytes(d, U, N));
    unsigned char *p = static_cast<unsigned char *>(workspace);
    uint4 *users_bf = reinterpret_cast<uint4 *>(p);  p += align_up((size_t)U * 4 * d, 256);
    uint4 *items_bf = reinterpret_cast<uint4 *>(p);  p += align_up((size_t)N * 4 * d, 256);
    float *unorm = reinterpret_cast<float *>(p);     p += align_up((size_t)U * 4, 256);
    uint32_t *qmax_bits = reinterpret_cast<uint32_t *>(p);
    fill_words(qmax_bits, 1, 0u, st);
    dim3 grid((N + 31) / 32, (U + 31) / 32);
    MACR_DISPATCH_LPR(d, {
        constexpr int D = 4 * LPR;
        k_bf16_prep<D><<<bf16_prep_blocks(U, N, D), 256, 0, st>>>(U, N, users, nullptr, items, users_bf, items_bf, unorm, qmax_bits);
        k_test_bf16_products<D><<<grid, 64, 0, st>>>(U, N, users_bf, items_bf, unorm, qmax_bits, c, prod, margin);
    });
    MACR_CHECK_LAUNCH("test_bf16_products", st);
    return MACR_OK;
}

extern "C" size_t macr_test_bf16_scores_workspace_bytes(int d, int U, int N) {
    if (!dim_supported(d) || U <= 0 || N <= 0) return 0;
    return align_up((size_t)U * 4 * d, 256) + align_up(items_c_bytes(N, d), 256) + align_up((size_t)U * 4, 256) + 256;
}

extern "C" int macr_test_bf16_scores(int score_kind, int d, int U, int N, const float *users, const float *items, const float *sig_u,
                                     const float *sig_i, float c, float *scores, float *margin, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(score_kind_valid(score_kind), MACR_E_INVALID, "test_bf16_scores: score_kind=%d", score_kind);
    MACR_REQUIRE(U > 0 && N > 0, MACR_E_INVALID, "test_bf16_scores: U=%d N=%d", U, N);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "test_bf16_scores: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(users && items && scores && margin && workspace, MACR_E_INVALID, "test_bf16_scores: null pointer");
    MACR_REQUIRE((!score_uses_sig_i(score_kind) || sig_i) && (!score_uses_sig_u(score_kind) || sig_u), MACR_E_INVALID,
                 "test_bf16_scores: score_kind %d needs its sigmoids", score_kind);
    MACR_REQUIRE(workspace_bytes >= macr_test_bf16_scores_workspace_bytes(d, U, N), MACR_E_WORKSPACE,
                 "test_bf16_scores: workspace %zu < %zu", workspace_bytes, macr_test_bf16_scores_workspace_bytes(d, U, N));
    unsigned char *p = static_cast<unsigned char *>(workspace);
    uint4 *users_c = reinterpret_cast<uint4 *>(p);   p += align_up((size_t)U * 4 * d, 256);
    uint4 *items_c = reinterpret_cast<uint4 *>(p);   p += align_up(items_c_bytes(N, d), 256);
    float *unorm = reinterpret_cast<float *>(p);     p += align_up((size_t)U * 4, 256);
    uint32_t *qmax_bits = reinterpret_cast<uint32_t *>(p);
    fill_words(qmax_bits, 1, 0u, st);
    dim3 grid((N + 31) / 32, (U + 31) / 32);
    MACR_DISPATCH_DK(d, score_kind, {
        k_bf16_prep_c<D, KIND><<<bf16_prep_c_blocks(U, N, D, kPrepTripsAlone), 256, 0, st>>>(U, N, users, nullptr, items, sig_u, sig_i, c, nullptr, users_c, items_c,
                                                                            unorm, qmax_bits);
        k_test_bf16_scores<D, KIND><<<grid, 64, 0, st>>>(U, N, users_c, items_c, unorm, qmax_bits, sig_u, c, scores, margin);
    });
    MACR_CHECK_LAUNCH("test_bf16_scores", st);
    return MACR_OK;
}

// the same for the fp16 filter (k_score_stream_h's sequence on the copies bf16_prep_c_block<.., HALF> writes; workspace as above)
extern "C" int macr_test_f16_scores(int score_kind, int d, int U, int N, const float *users, const float *items, const float *sig_u,
                                    const float *sig_i, float c, float *scores, float *margin, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(score_kind_valid(score_kind), MACR_E_INVALID, "test_f16_scores: score_kind=%d", score_kind);
    MACR_REQUIRE(U > 0 && N > 0, MACR_E_INVALID, "test_f16_scores: U=%d N=%d", U, N);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "test_f16_scores: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(users && items && scores && margin && workspace, MACR_E_INVALID, "test_f16_scores: null pointer");
    MACR_REQUIRE((!score_uses_sig_i(score_kind) || sig_i) && (!score_uses_sig_u(score_kind) || sig_u), MACR_E_INVALID,
                 "test_f16_scores: score_kind %d needs its sigmoids", score_kind);
    MACR_REQUIRE(workspace_bytes >= macr_test_bf16_scores_workspace_bytes(d, U, N), MACR_E_WORKSPACE,
                 "test_f16_scores: workspace %zu < %zu", workspace_bytes, macr_test_bf16_scores_workspace_bytes(d, U, N));
    unsigned char *p = static_cast<unsigned char *>(workspace);
    uint4 *users_c = reinterpret_cast<uint4 *>(p);   p += align_up((size_t)U * 4 * d, 256);
    uint4 *items_c = reinterpret_cast<uint4 *>(p);   p += align_up(items_c_bytes(N, d), 256);
    float *unorm = reinterpret_cast<float *>(p);     p += align_up((size_t)U * 4, 256);
    uint32_t *qmax_bits = reinterpret_cast<uint32_t *>(p);
    fill_words(qmax_bits, 1, 0u, st);
    dim3 grid((N + 31) / 32, (U + 31) / 32);
    MACR_DISPATCH_DK(d, score_kind, {
        k_bf16_prep_c<D, KIND, true><<<bf16_prep_c_blocks(U, N, D, kPrepTripsAlone), 256, 0, st>>>(U, N, users, nullptr, items, sig_u, sig_i, c, nullptr, users_c,
                                                                                                 items_c, unorm, qmax_bits);
        k_test_bf16_scores<D, KIND, true><<<grid, 64, 0, st>>>(U, N, users_c, items_c, unorm, qmax_bits, sig_u, c, scores, margin);
    });
    MACR_CHECK_LAUNCH("test_f16_scores", st);
    return MACR_OK;
}
#endif  // MACR_TEST_ENTRY_POINTS

extern "C" size_t macr_topk_scores_workspace_bytes(int rows, int K) {
    return (rows > 0 && K > MACR_MAX_TOPK) ? align_up((size_t)rows * sizeof(uint64_t), 256) : 0;
}

extern "C" int macr_topk_scores(const float *scores, int cols, int rows, int K, int32_t *out_idx,
                                float *out_val, void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(scores && out_idx, MACR_E_INVALID, "topk_scores: null pointer");
    MACR_REQUIRE(cols > 0 && rows > 0, MACR_E_INVALID, "topk_scores: cols=%d rows=%d", cols, rows);
    MACR_REQUIRE(K >= 1, MACR_E_INVALID, "topk_scores: K=%d", K);
    if (K <= MACR_MAX_TOPK_FUSED) k_topk_scores<<<(rows + 3) / 4, 256, 0, st>>>(scores, cols, rows, K, out_idx, out_val);
    else if (K <= MACR_MAX_TOPK) k_topk_scores_wide<<<(rows + 3) / 4, 256, 0, st>>>(scores, cols, rows, K, out_idx, out_val, nullptr, 0, 0, 0, K, 0, nullptr, nullptr);
    else {
        // c_top_k_array_index takes any top_k (tools.h:13-22): rounds of 128 positions, each bounded by the last key of the one before
        MACR_REQUIRE(workspace && workspace_bytes >= macr_topk_scores_workspace_bytes(rows, K), MACR_E_WORKSPACE,
                     "topk_scores: K=%d > %d needs a workspace of %zu bytes (macr_topk_scores_workspace_bytes)", K, MACR_MAX_TOPK,
                     macr_topk_scores_workspace_bytes(rows, K));
        uint64_t *last = static_cast<uint64_t *>(workspace);
        for (int off = 0; off < K; off += MACR_MAX_TOPK) {
            const int k = std::min(MACR_MAX_TOPK, K - off);
            k_topk_scores_wide<<<(rows + 3) / 4, 256, 0, st>>>(scores, cols, rows, k, out_idx, out_val, nullptr, 0, 0, 0, K, off,
                                                                off ? last : nullptr, last);
            MACR_CHECK_LAUNCH("topk_scores", st);
        }
    }
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
    if (K <= MACR_MAX_TOPK_FUSED) k_topk_merge<<<(U + 3) / 4, 256, 0, st>>>(W, U, K, vals, idxs, fill_mask_ptr, fill_mask_idx, out_val, out_idx, out_cnt);
    else k_topk_merge_wide<<<(U + 3) / 4, 256, 0, st>>>(W, U, K, vals, idxs, fill_mask_ptr, fill_mask_idx, out_val, out_idx, out_cnt);
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

extern "C" int macr_branch_sigmoid2(int d, const float *rows_a, const int32_t *idx_a, int n_a, const float *w_a, float *out_a,
                                    const float *rows_b, const int32_t *idx_b, int n_b, const float *w_b, float *out_b, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(rows_a && w_a && out_a && rows_b && w_b && out_b, MACR_E_INVALID, "branch_sigmoid2: null pointer");
    MACR_REQUIRE(n_a > 0 && n_b > 0, MACR_E_INVALID, "branch_sigmoid2: n=%d, %d", n_a, n_b);
    MACR_REQUIRE(dim_supported(d), MACR_E_UNSUPPORTED, "branch_sigmoid2: d=%d not in {32,64,128,256}", d);
    MACR_DISPATCH_LPR(d, {
        const int rpb = 256 / LPR, nb_a = (n_a + rpb - 1) / rpb, nb_b = (n_b + rpb - 1) / rpb;
        k_branch_sigmoid2<LPR><<<nb_a + nb_b, 256, 0, st>>>(nb_a, rows_a, idx_a, n_a, w_a, out_a, rows_b, idx_b, n_b, w_b, out_b);
    });
    MACR_CHECK_LAUNCH("branch_sigmoid", st);
    return MACR_OK;
}

extern "C" int macr_metrics_foldout(int U, int K, const int32_t *rankings, const int32_t *gt_ptr,
                                    const int32_t *gt_idx, float *results, int hr_in_ap_slot, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && K >= 1, MACR_E_INVALID, "metrics_foldout: U=%d K=%d", U, K);
    MACR_REQUIRE(rankings && gt_ptr && gt_idx && results, MACR_E_INVALID, "metrics_foldout: null pointer");
    if (K <= 128) k_metrics_foldout_w<<<(U + 3) / 4, 256, 0, st>>>(U, K, rankings, nullptr, nullptr, gt_ptr, gt_idx, results, hr_in_ap_slot);
    else k_metrics_foldout<<<(U + 127) / 128, 128, 0, st>>>(U, K, rankings, gt_ptr, gt_idx, results, hr_in_ap_slot);
    MACR_CHECK_LAUNCH("metrics_foldout", st);
    return MACR_OK;
}

extern "C" int macr_metrics_foldout_fill(int U, int K, const int32_t *rankings, const int32_t *fill_ptr, const int32_t *fill_idx,
                                         const int32_t *gt_ptr, const int32_t *gt_idx, float *results, int hr_in_ap_slot,
                                         void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && K >= 1 && K <= MACR_MAX_TOPK, MACR_E_INVALID, "metrics_foldout_fill: U=%d K=%d (K <= %d)", U, K, MACR_MAX_TOPK);
    MACR_REQUIRE(rankings && fill_ptr && fill_idx && gt_ptr && gt_idx && results, MACR_E_INVALID, "metrics_foldout_fill: null pointer");
    k_metrics_foldout_w<<<(U + 3) / 4, 256, 0, st>>>(U, K, rankings, fill_ptr, fill_idx, gt_ptr, gt_idx, results, hr_in_ap_slot);
    MACR_CHECK_LAUNCH("metrics_foldout", st);
    return MACR_OK;
}

extern "C" int macr_metrics_mf(int U, int Kmax, const int32_t *rankings, const int32_t *cnt,
                               const int32_t *gt_ptr, const int32_t *gt_idx, const int32_t *Ks, int nK,
                               double *out, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && Kmax >= 1, MACR_E_INVALID, "metrics_mf: U=%d Kmax=%d", U, Kmax);
    MACR_REQUIRE(Kmax <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "metrics_mf: Kmax=%d > %d", Kmax, MACR_MAX_TOPK);
    MACR_REQUIRE(rankings && gt_ptr && gt_idx && Ks && out, MACR_E_INVALID, "metrics_mf: null pointer");
    MACR_REQUIRE(nK >= 1 && nK <= 8, MACR_E_UNSUPPORTED, "metrics_mf: nK=%d outside [1,8]", nK);
    KsArg ka;
    ka.n = nK;
    for (int q = 0; q < nK; ++q) {
        MACR_REQUIRE(Ks[q] >= 1 && Ks[q] <= Kmax, MACR_E_INVALID, "metrics_mf: Ks[%d]=%d outside [1,%d]", q, Ks[q], Kmax);
        ka.k[q] = Ks[q];
    }
    k_metrics_mf<<<(U + 3) / 4, 256, 0, st>>>(U, Kmax, rankings, cnt, gt_ptr, gt_idx, ka, out);
    MACR_CHECK_LAUNCH("metrics_mf", st);
    return MACR_OK;
}

extern "C" size_t macr_metrics_mf_mean_workspace_bytes(int U, int nK) {
    if (U <= 0 || nK < 1 || nK > 8) return 0;
    return 256 + (size_t)((U + kMeanPerBlock - 1) / kMeanPerBlock) * 4 * nK * sizeof(double);
}

extern "C" int macr_metrics_mf_mean(int U, int Kmax, const int32_t *rankings, const int32_t *cnt,
                                    const int32_t *gt_ptr, const int32_t *gt_idx, const int32_t *Ks, int nK,
                                    double *per_user, double *mean, void *workspace, size_t workspace_bytes, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(U > 0 && Kmax >= 1, MACR_E_INVALID, "metrics_mf_mean: U=%d Kmax=%d", U, Kmax);
    MACR_REQUIRE(Kmax <= MACR_MAX_TOPK, MACR_E_UNSUPPORTED, "metrics_mf_mean: Kmax=%d > %d", Kmax, MACR_MAX_TOPK);
    MACR_REQUIRE(rankings && gt_ptr && gt_idx && Ks && mean, MACR_E_INVALID, "metrics_mf_mean: null pointer");
    MACR_REQUIRE(nK >= 1 && nK <= 8, MACR_E_UNSUPPORTED, "metrics_mf_mean: nK=%d outside [1,8]", nK);
    MACR_REQUIRE(workspace && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0, MACR_E_INVALID,
                 "metrics_mf_mean: workspace is null or not 8-byte aligned (macr_metrics_mf_mean_workspace_bytes)");
    MACR_REQUIRE(workspace_bytes >= macr_metrics_mf_mean_workspace_bytes(U, nK), MACR_E_WORKSPACE,
                 "metrics_mf_mean: workspace %zu < %zu bytes", workspace_bytes, macr_metrics_mf_mean_workspace_bytes(U, nK));
    KsArg ka;
    ka.n = nK;
    for (int q = 0; q < nK; ++q) {
        MACR_REQUIRE(Ks[q] >= 1 && Ks[q] <= Kmax, MACR_E_INVALID, "metrics_mf_mean: Ks[%d]=%d outside [1,%d]", q, Ks[q], Kmax);
        ka.k[q] = Ks[q];
    }
    int32_t *ticket = static_cast<int32_t *>(workspace);
    double *partials = reinterpret_cast<double *>(static_cast<char *>(workspace) + 256);
    k_metrics_mf_mean<<<(U + kMeanPerBlock - 1) / kMeanPerBlock, 64 * kMeanWaves, 0, st>>>(U, Kmax, rankings, cnt, gt_ptr, gt_idx, ka, per_user,
                                                                                     partials, ticket, mean);
    MACR_CHECK_LAUNCH("metrics_mf_mean", st);
    return MACR_OK;
}

extern "C" int macr_colmean(const void *in, int in_is_f32, int rows, int cols, double *out, void *stream) {
    hipStream_t st = as_stream(stream);
    MACR_REQUIRE(in && out && rows > 0 && cols > 0, MACR_E_INVALID, "colmean: bad arguments");
    // (wider blocks were tried for wide matrices -- 32 or 8 columns each, coalesced row pieces: 46 and 23 us for 15 424 x 100
    // floats against 18 with a block per column: too few blocks)
    if (in_is_f32)
        k_colmean<float><<<cols, 1024, 0, st>>>(static_cast<const float *>(in), rows, cols, out);
    else
        k_colmean<double><<<cols, 1024, 0, st>>>(static_cast<const double *>(in), rows, cols, out);
    MACR_CHECK_LAUNCH("colmean", st);
    return MACR_OK;
}
#endif  // MACR_EVAL_KERNELS_ONLY
