// Device-side (u, i+, i-) sampler -- SURVEY.md 8(f2): removes the host sampler ceiling
// (the reference's Data.sample runs at ~0.5 M triples/s in Python; a training step takes ~60 us).
//
// Same DISTRIBUTION as the reference samplers (macr_mf/load_data.py:543-566,
// macr_lightgcn/utility/load_data.py:174-212), not the same random stream (that needs Python's
// Mersenne Twister; use --sampler reference for stream-exact runs):
//   users : B distinct users drawn uniformly from the pool (rd.sample), or B independent draws
//           when B exceeds the pool (rd.choice);
//   pos   : uniform over the user's train list (item 0 when the list is empty, load_data.py:551-552);
//   neg   : uniform over the items NOT in the user's train list (rejection sampling, :554-558).
// Counter-based randomness: every draw is a hash of (seed, step, triple, draw number), so a batch is a
// pure function of (seed, step) -- reproducible and order-independent.  Sampling without replacement
// takes the first B images of a keyed pseudo-random permutation of the pool (4-round Feistel network
// over the next power of two with cycle walking).
#include "common.hpp"

namespace macr {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {          // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t draw(uint64_t key, uint32_t t, uint32_t n) {
    return (uint32_t)(mix64(key ^ ((uint64_t)t << 32 | n)) >> 32);
}
// unbiased-enough uniform integer in [0, range): 32x32->64 multiply-shift (bias < range / 2^32)
__device__ __forceinline__ uint32_t below(uint32_t r, uint32_t range) { return (uint32_t)(((uint64_t)r * range) >> 32); }

__device__ __forceinline__ uint32_t feistel_perm(uint32_t x, uint32_t n, int bits, uint64_t key) {
    const int hb = (bits + 1) / 2;                  // half width (the domain is 2^(2*hb) >= n)
    const uint32_t mask = (1u << hb) - 1u;
    do {
        uint32_t l = x >> hb, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = (uint32_t)(mix64(key + 0x1234567ull * (round + 1) + r) >> 17) & mask;
            const uint32_t nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = (l << hb) | r;
    } while (x >= n);                               // cycle walking keeps it a bijection on [0, n)
    return x;
}

__global__ __launch_bounds__(256) void k_sample_triples(uint64_t seed, uint64_t step, int B, int n_items,
                                                        const int32_t *__restrict__ pool, int n_pool, int pool_bits,
                                                        const int32_t *__restrict__ train_ptr,
                                                        const int32_t *__restrict__ train_idx,
                                                        const int32_t *__restrict__ excl_ptr,
                                                        const int32_t *__restrict__ excl_idx,
                                                        int32_t *__restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B) return;
    step += blockIdx.y;                                   // macr_sample_triples_many: one grid row per step
    out += (size_t)blockIdx.y * 3 * B;
    const uint64_t key = mix64(seed * 0x9e3779b97f4a7c15ull + step);
    uint32_t slot;
    if (B <= n_pool) slot = feistel_perm((uint32_t)t, (uint32_t)n_pool, pool_bits, key);
    else slot = below(draw(key, t, 0), (uint32_t)n_pool);
    const int user = pool ? pool[slot] : (int)slot;
    const int beg = train_ptr[user], len = train_ptr[user + 1] - beg;
    const int pos = len > 0 ? train_idx[beg + (int)below(draw(key, t, 1), (uint32_t)len)] : 0;
    // negatives avoid the user's exclusion list: the list the positive came from, or a second one (the test-set sampler of
    // LightGCN excludes test AND train items, utility/load_data.py:233-239)
    const int32_t *xi = excl_ptr ? excl_idx : train_idx;
    const int xbeg = excl_ptr ? excl_ptr[user] : beg, xlen = excl_ptr ? excl_ptr[user + 1] - xbeg : len;
    int neg = 0;
    for (uint32_t n = 2; n < 2 + 4096; ++n) {       // bounded: a user owning (almost) the whole catalogue cannot hang the GPU
        neg = (int)below(draw(key, t, n), (uint32_t)n_items);
        int lo = 0, hi = xlen;                      // binary search in the ascending list
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (xi[xbeg + mid] < neg) lo = mid + 1; else hi = mid; }
        if (!(lo < xlen && xi[xbeg + lo] == neg)) break;
    }
    out[t] = user; out[B + t] = pos; out[2 * (size_t)B + t] = neg;
}

}  // namespace macr

extern "C" int macr_sample_triples(uint64_t seed, uint64_t step, int B, int n_items, const int32_t *pool, int n_pool,
                                   const int32_t *train_ptr, const int32_t *train_idx, int32_t *out, void *stream) {
    using namespace macr;
    MACR_REQUIRE(B > 0 && n_items > 0 && n_pool > 0, MACR_E_INVALID, "sample_triples: B=%d n_items=%d n_pool=%d", B, n_items, n_pool);
    MACR_REQUIRE(train_ptr && train_idx && out, MACR_E_INVALID, "sample_triples: null pointer");
    int bits = 1;
    while ((1u << bits) < (unsigned)n_pool) ++bits;
    if (bits & 1) ++bits;                           // even width: two equal Feistel halves
    hipStream_t st = as_stream(stream);
    k_sample_triples<<<(B + 255) / 256, 256, 0, st>>>(seed, step, B, n_items, pool, n_pool, bits, train_ptr, train_idx, nullptr, nullptr, out);
    MACR_CHECK_LAUNCH("sample_triples", st);
    return MACR_OK;
}

extern "C" int macr_sample_triples_many(uint64_t seed, uint64_t step0, int n_steps, int B, int n_items, const int32_t *pool,
                                        int n_pool, const int32_t *train_ptr, const int32_t *train_idx,
                                        const int32_t *excl_ptr, const int32_t *excl_idx, int32_t *out, void *stream) {
    using namespace macr;
    MACR_REQUIRE(B > 0 && n_items > 0 && n_pool > 0 && n_steps > 0 && n_steps <= 65535, MACR_E_INVALID,
                 "sample_triples_many: B=%d n_items=%d n_pool=%d n_steps=%d", B, n_items, n_pool, n_steps);
    MACR_REQUIRE(train_ptr && train_idx && out, MACR_E_INVALID, "sample_triples_many: null pointer");
    MACR_REQUIRE((excl_ptr == nullptr) == (excl_idx == nullptr), MACR_E_INVALID, "sample_triples_many: excl_ptr without excl_idx");
    int bits = 1;
    while ((1u << bits) < (unsigned)n_pool) ++bits;
    if (bits & 1) ++bits;
    hipStream_t st = as_stream(stream);
    k_sample_triples<<<dim3((B + 255) / 256, n_steps), 256, 0, st>>>(seed, step0, B, n_items, pool, n_pool, bits, train_ptr,
                                                                      train_idx, excl_ptr, excl_idx, out);
    MACR_CHECK_LAUNCH("sample_triples", st);
    return MACR_OK;
}
