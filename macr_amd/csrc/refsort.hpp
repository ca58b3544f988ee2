// Stable LSD radix sort of (key, value) pairs of 32-bit words, for the row references of a training batch.
//
// Why a sort at all: the gradient of a batch is a sum over its 3B row references (macr_mf/model.py:35-37 gathers,
// :74 IndexedSlices de-duplication before the sparse Adam apply), and the reference's sampler draws positives by
// popularity (macr_mf/load_data.py:543-566), so a batch references a handful of rows hundreds of times.  Ordering
// the references by row turns "hundreds of atomics on one L2 line" into segments:
//   * small batches (B <= kSmallBatchMax): ONE workgroup buckets the triples of the batch by the low byte of the
//     positive item while the (B,B) kernel runs (batch_bucket_block in train_kernels.hip, one pass built from the
//     pieces below); pair_bwd then adds equal positive rows of a 16-slot chunk once;
//   * large batches: all 3B references are sorted by row with the multi-block kernels below and a segment-reduce
//     pass sums each row's contributions from a staging buffer written with plain stores -- no atomics on the path.
// 8-bit digits; a wave ranks 64 keys at a time with ballots (no LDS atomics in the ranking step); every pass is
// stable, so equal rows keep their batch order and the summation order of a row is the same in every run.
#pragma once
#include <algorithm>
#include "common.hpp"

namespace macr {

constexpr int kRadix = 256;
constexpr int kSortTile = 16384;         // keys per workgroup in the multi-block passes (16 waves x 1024)

// Lanes of the wave that hold the same 8-bit digit as this lane (valid lanes only).
__device__ __forceinline__ uint64_t match_digit(uint32_t dgt, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (dgt >> b) & 1u;
        const uint64_t m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

constexpr int kSortBatch = 8;            // keys per lane fetched before any is ranked (the loops are latency bound)

// hist[digit] += number of keys of [lo,hi) with that digit (hist: 256 LDS words owned by this wave, zeroed).
// Loads use clamped addresses instead of predicates: a predicated load compiles to a branch with its own wait, which
// serialises the batch (measured: 8 us for 8 loads).
__device__ __forceinline__ void wave_count(const uint32_t *__restrict__ kin, int lo, int hi, int shift, uint32_t *hist) {
    const int lane = threadIdx.x & 63;
    for (int base = lo; base < hi; base += 64 * kSortBatch) {
        uint32_t k[kSortBatch];
#pragma unroll
        for (int q = 0; q < kSortBatch; ++q) { const int p = base + q * 64 + lane; k[q] = kin[p < hi ? p : hi - 1]; }
#pragma unroll
        for (int q = 0; q < kSortBatch; ++q)
            if (base + q * 64 + lane < hi) atomicAdd(&hist[(k[q] >> shift) & 255u], 1u);
    }
}

// The wave moves keys [lo,hi) to kout/vout in order, 64 at a time: offs[digit] is the next free output position of
// that digit for THIS wave (LDS, owned by the wave; advanced here: LDS operations of a wave execute in order).
__device__ __forceinline__ void wave_rank_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                  int lo, int hi, int shift, uint32_t *offs,
                                                  uint32_t *__restrict__ kout, uint32_t *__restrict__ vout) {
    const int lane = threadIdx.x & 63;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int base = lo; base < hi; base += 64 * kSortBatch) {
        uint32_t kk[kSortBatch], vv[kSortBatch];
#pragma unroll
        for (int q = 0; q < kSortBatch; ++q) {
            const int p = base + q * 64 + lane, pc = p < hi ? p : hi - 1;
            kk[q] = kin[pc];
            vv[q] = vin[pc];
        }
#pragma unroll
        for (int q = 0; q < kSortBatch; ++q) {
            if (base + q * 64 >= hi) break;                               // wave-uniform
            const bool valid = base + q * 64 + lane < hi;
            const uint32_t dgt = (kk[q] >> shift) & 255u;
            const uint64_t peers = match_digit(dgt, valid);
            const uint32_t rank = (uint32_t)__popcll(peers & below);
            const uint32_t dst = offs[dgt] + rank;
            if (valid && rank == 0) offs[dgt] = dst + (uint32_t)__popcll(peers);     // lowest peer advances the digit
            if (valid) { kout[dst] = kk[q]; vout[dst] = vv[q]; }
        }
    }
}

// Turns per-wave digit counts s_hist[w][digit] into output positions: position of (digit, wave) in digit-major,
// wave-minor order, plus gbase[digit] (the digit's start outside this workgroup; NULL = exclusive scan over digits).
// All threads of the block must call it (contains barriers).  s_wtot: 4 LDS words.
template <int NW>
__device__ __forceinline__ void block_digit_offsets(uint32_t *s_hist, uint32_t *s_wtot, const uint32_t *gbase, int gstride) {
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    uint32_t tot = 0, excl = 0;
    if (t < kRadix) {
#pragma unroll 4
        for (int w = 0; w < NW; ++w) { const uint32_t c = s_hist[w * kRadix + t]; s_hist[w * kRadix + t] = tot; tot += c; }
        if (gbase) {
            excl = gbase[(size_t)t * gstride];
        } else {
            uint32_t inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o, kWave); if (lane >= o) inc += x; }
            excl = inc - tot;
            if (lane == 63) s_wtot[wid] = inc;
        }
    }
    __syncthreads();
    if (t < kRadix) {
        if (!gbase)
            for (int w = 0; w < wid; ++w) excl += s_wtot[w];
#pragma unroll 4
        for (int w = 0; w < NW; ++w) s_hist[w * kRadix + t] += excl;
    }
    __syncthreads();
}

// ---- multi-block passes (large batches) -------------------------------------------------------------------------
// A workgroup of 16 waves owns a tile of TILE keys (a wave's share is a chain of memory and LDS latencies, so it is kept
// short).  Per pass:
//   k_rs_count    ghist[block][digit] = the tile's digit counts; gtot[digit] += them (256 atomics per workgroup)
//   k_rs_scatter  every workgroup derives its own output positions -- start of the digit (exclusive scan of gtot over
//                 the digits) + the counts of the tiles before it (a column of ghist: block-major, so the 256 digit
//                 threads read coalesced rows) -- which replaces a one-workgroup scan of all 256 x nblk counts between
//                 the two kernels (13 us per pass at 3 M keys).  The tile is then ordered by digit IN LDS (per-wave
//                 ballot ranking, stable) and written out position by position: keys of one digit leave as one
//                 contiguous run instead of 64 four-byte stores to 64 places per wave instruction (PMC WRITE_SIZE of
//                 the direct scatter: 110 MB for the 24 MB of pairs of a 3 M-key pass).
template <int TILE>
static __global__ __launch_bounds__(1024) void k_rs_count(const uint32_t *__restrict__ kin, int n, int shift,
                                                          uint32_t *__restrict__ ghist, uint32_t *__restrict__ gtot) {
    __shared__ uint32_t s_hist[kRadix];
    if (threadIdx.x < kRadix) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int lo = blockIdx.x * TILE, hi = lo + TILE < n ? lo + TILE : n;
    constexpr int NQ = TILE / 1024;
    uint32_t k[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int p = lo + q * 1024 + threadIdx.x; k[q] = kin[p < hi ? p : hi - 1]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (lo + q * 1024 + (int)threadIdx.x < hi) atomicAdd(&s_hist[(k[q] >> shift) & 255u], 1u);
    __syncthreads();
    if (threadIdx.x < kRadix) {
        const uint32_t c = s_hist[threadIdx.x];
        ghist[(size_t)blockIdx.x * kRadix + threadIdx.x] = c;
        if (c) atomicAdd(&gtot[threadIdx.x], c);
    }
}

// The wave ranks keys [lo,hi) and stores them at their position inside the tile's digit order (LDS): offs[digit] is the
// next free position of that digit for THIS wave (LDS, owned by the wave; LDS operations of a wave execute in order).
__device__ __forceinline__ void wave_rank_to_lds(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                 int lo, int hi, int shift, uint32_t *offs, uint32_t *s_k, uint32_t *s_v) {
    const int lane = threadIdx.x & 63;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int base = lo; base < hi; base += 64 * kSortBatch) {
        uint32_t kk[kSortBatch], vv[kSortBatch];
#pragma unroll
        for (int q = 0; q < kSortBatch; ++q) {
            const int p = base + q * 64 + lane, pc = p < hi ? p : hi - 1;
            kk[q] = kin[pc];
            vv[q] = vin[pc];
        }
#pragma unroll
        for (int q = 0; q < kSortBatch; ++q) {
            if (base + q * 64 >= hi) break;                               // wave-uniform
            const bool valid = base + q * 64 + lane < hi;
            const uint32_t dgt = (kk[q] >> shift) & 255u;
            const uint64_t peers = match_digit(dgt, valid);
            const uint32_t rank = (uint32_t)__popcll(peers & below);
            const uint32_t dst = offs[dgt] + rank;
            if (valid && rank == 0) offs[dgt] = dst + (uint32_t)__popcll(peers);     // lowest peer advances the digit
            if (valid) { s_k[dst] = kk[q]; s_v[dst] = vv[q]; }
        }
    }
}

// dynamic LDS: 2 * TILE words (the tile's keys and values in digit order)
template <int TILE>
static __global__ __launch_bounds__(1024) void k_rs_scatter(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                            uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, int n,
                                                            int shift, const uint32_t *__restrict__ ghist,
                                                            const uint32_t *__restrict__ gtot) {
    extern __shared__ uint32_t s_tile[];
    __shared__ uint32_t s_hist[16 * kRadix];
    __shared__ uint32_t s_base[kRadix];                 // global position of a digit's first key of this tile - its position in the tile
    __shared__ uint32_t s_wtot[2][4];
    uint32_t *s_k = s_tile, *s_v = s_tile + TILE;
    const int t = threadIdx.x, wid = t >> 6, lane = t & 63;
    constexpr int per = TILE / 16;
    const int blo = blockIdx.x * TILE, bhi = blo + TILE < n ? blo + TILE : n;
    const int lo = blo + wid * per < bhi ? blo + wid * per : bhi;
    const int hi = lo + per < bhi ? lo + per : bhi;
    for (int k = t; k < 16 * kRadix; k += 1024) s_hist[k] = 0;
    // (issued before the counting loop: the column of earlier tiles' counts and the digit totals)
    uint32_t before = 0, gt = 0;
    if (t < kRadix) {
        gt = gtot[t];
        uint32_t acc[4] = {0, 0, 0, 0};
        int bb = 0;
        for (; bb + 4 <= (int)blockIdx.x; bb += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += ghist[(size_t)(bb + q) * kRadix + t];
        }
        for (; bb < (int)blockIdx.x; ++bb) acc[0] += ghist[(size_t)bb * kRadix + t];
        before = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    __syncthreads();
    wave_count(kin, lo, hi, shift, s_hist + wid * kRadix);
    __syncthreads();
    uint32_t tot = 0, inc_l = 0, inc_g = 0;
    if (t < kRadix) {
#pragma unroll 4
        for (int w = 0; w < 16; ++w) { const uint32_t c = s_hist[w * kRadix + t]; s_hist[w * kRadix + t] = tot; tot += c; }
        inc_l = tot; inc_g = gt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t x = __shfl_up(inc_l, o, kWave), y = __shfl_up(inc_g, o, kWave);
            if (lane >= o) { inc_l += x; inc_g += y; }
        }
        if (lane == 63) { s_wtot[0][wid] = inc_l; s_wtot[1][wid] = inc_g; }
    }
    __syncthreads();
    if (t < kRadix) {
        uint32_t start_l = inc_l - tot, start_g = inc_g - gt;
        for (int w = 0; w < wid; ++w) { start_l += s_wtot[0][w]; start_g += s_wtot[1][w]; }
        s_base[t] = start_g + before - start_l;
#pragma unroll 4
        for (int w = 0; w < 16; ++w) s_hist[w * kRadix + t] += start_l;
    }
    __syncthreads();
    wave_rank_to_lds(kin, vin, lo, hi, shift, s_hist + wid * kRadix, s_k, s_v);
    __syncthreads();
    const int cnt = bhi - blo;
    for (int q = t; q < cnt; q += 1024) {
        const uint32_t key = s_k[q];
        const uint32_t g = s_base[(key >> shift) & 255u] + (uint32_t)q;
        kout[g] = key;
        vout[g] = s_v[q];
    }
}

static inline int key_bits_for(uint32_t max_key) {
    int b = 1;
    while (b < 32 && (max_key >> b)) ++b;
    return b;
}

// Small arrays take 2048-key tiles: with 16384-key tiles the 24 576 references of a B = 8192 batch are two workgroups
// (two CUs) per pass, ~30 us each; twelve workgroups bring a pass to the launch-bound floor.
constexpr int kSortTileSmall = 2048, kSortSmallMax = 1 << 17;
// In between, 8192-key tiles (measured, ref_sort of 3B references: B = 2^16 98 -> 69 us, 2^18 109 -> 88 us; 2^20 176 -> 207 us)
constexpr int kSortTileMid = 8192, kSortMidMax = 1 << 20;
static inline int sort_tile_for(int n) {
    static const char *env = getenv("MACR_SORT_TILE");          // measurements: 2048 | 8192 | 16384 for every size
    if (env) { const int t = atoi(env); if (t == kSortTileSmall || t == kSortTileMid || t == kSortTile) return t; }
    return n <= kSortSmallMax ? kSortTileSmall : n <= kSortMidMax ? kSortTileMid : kSortTile;
}
constexpr int kSortMaxPasses = 4;
// Words of tile counts + digit totals a sort of UP TO n pairs may need: monotone in n although the tile size is not
// (n = 2^17 sorts in 64 tiles of 2048 keys, n = 2^17 + 1 in 17 of 8192), so a workspace sized for a batch capacity
// serves every smaller batch (a MACR_SORT_TILE override counts as the small tile: the largest count).
static inline size_t sort_hist_words(int n) {
    size_t tiles = 0;
    const int small_n = n < kSortSmallMax ? n : kSortSmallMax, mid_n = n < kSortMidMax ? n : kSortMidMax;
    tiles = (size_t)(small_n + kSortTileSmall - 1) / kSortTileSmall;
    if (n > kSortSmallMax) tiles = std::max(tiles, (size_t)(mid_n + kSortTileMid - 1) / kSortTileMid);
    if (n > kSortMidMax) tiles = std::max(tiles, (size_t)((size_t)n + kSortTile - 1) / kSortTile);
    if (getenv("MACR_SORT_TILE")) tiles = std::max(tiles, ((size_t)n + kSortTileSmall - 1) / kSortTileSmall);
    return (size_t)kRadix * tiles + (size_t)kSortMaxPasses * kRadix;
}

template <int TILE>
static inline void launch_rs_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int n, int shift,
                                  uint32_t *ghist, uint32_t *gtot, int nblk, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rs_scatter<TILE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE * 4);     // (a failure shows at the launch)
        attr_set = true;
    }
    k_rs_count<TILE><<<nblk, 1024, 0, st>>>(kin, n, shift, ghist, gtot);
    k_rs_scatter<TILE><<<nblk, 1024, 2 * TILE * 4, st>>>(kin, vin, kout, vout, n, shift, ghist, gtot);
}

// Sorts n pairs held in (ka,va) with (kb,vb) as the second buffer; returns 0 if the result is in (ka,va), 1 if in
// (kb,vb).  ghist: sort_hist_words(n) words.
static inline int launch_radix_sort(uint32_t *ka, uint32_t *va, uint32_t *kb, uint32_t *vb, int n, uint32_t max_key,
                                    uint32_t *ghist, hipStream_t st) {
    const int tile = sort_tile_for(n);
    const int nblk = (n + tile - 1) / tile, bits = key_bits_for(max_key);
    uint32_t *gtot = ghist + (size_t)kRadix * nblk;
    fill_words(gtot, (size_t)kSortMaxPasses * kRadix, 0u, st);
    int flip = 0, pass = 0;
    for (int shift = 0; shift < bits; shift += 8, ++pass) {
        uint32_t *kin = flip ? kb : ka, *vin = flip ? vb : va, *kout = flip ? ka : kb, *vout = flip ? va : vb;
        uint32_t *gt = gtot + (size_t)pass * kRadix;
        if (tile == kSortTileSmall) launch_rs_pass<kSortTileSmall>(kin, vin, kout, vout, n, shift, ghist, gt, nblk, st);
        else if (tile == kSortTileMid) launch_rs_pass<kSortTileMid>(kin, vin, kout, vout, n, shift, ghist, gt, nblk, st);
        else launch_rs_pass<kSortTile>(kin, vin, kout, vout, n, shift, ghist, gt, nblk, st);
        flip ^= 1;
    }
    return flip;
}

}  // namespace macr
