// Stand-alone bench of the evaluator's bf16 listing pass (eval_kernels.hip k_score_stream_b) and of candidate rewrites of it:
// builds in seconds (the kernel templates of eval_kernels.hip without its C ABI), runs every variant on the same operands and
// thresholds, and checks that the variants list the same number of candidates.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include tools/listing_bench.hip -o tools/listing_bench
//   tools/listing_bench [U] [N] [candidates per user] [reps]
#define MACR_EVAL_KERNELS_ONLY
#include "../macr_amd/csrc/eval_kernels.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace macr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Rewrite: the epilogue inside the matrix product.
//   items:  q'_i = sig_i * q_i (RUBI kinds), two-term bf16 split as before, plus ONE extra k-slab per item that holds the
//           three-term bf16 split of the bias  -c * sig_i  against a user-side slab of ones
//   acc''   = sum_k u_k q'_ik - c sig_i      straight out of the MFMAs: the listing test is  acc'' >= tau / sig_u  (a lane
//           constant: no per-item filter value, no fma per score, no staging of sig_i), a listed score is acc'' * sig_u.
// The bias slab is the LAST MFMA: the products accumulate at their own (small) magnitude and the bias costs one rounding at
// the magnitude of c, like the (acc - c) of the fp32 epilogue.
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_prep_items2(int n, const float *__restrict__ items, const float *__restrict__ sig_i, float c,
                                                     uint4 *__restrict__ items_bf, uint4 *__restrict__ items_ext) {
    constexpr int LPRB = D / 8;
    const int sub = threadIdx.x % LPRB, row = blockIdx.x * (256 / LPRB) + threadIdx.x / LPRB;
    if (row >= n) return;
    const float s = sig_i[row];
    float4 a = ld4(items + (size_t)row * D + 8 * sub), b = ld4(items + (size_t)row * D + 8 * sub + 4);
    const float x[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        hi[k] = bf16_rne_bits(x[k]);
        lo[k] = bf16_rne_bits(x[k] - __uint_as_float(hi[k] << 16));
    }
    uint4 *dst = items_bf + (size_t)row * 2 * LPRB;
    dst[sub] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    dst[LPRB + sub] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    if (sub == 0) {
        const float bias = -c * s;
        const uint32_t b1 = bf16_rne_bits(bias);
        const float r1 = bias - __uint_as_float(b1 << 16);
        const uint32_t b2 = bf16_rne_bits(r1);
        const uint32_t b3 = bf16_rne_bits(r1 - __uint_as_float(b2 << 16));
        items_ext[row] = make_uint4(b1 | (b2 << 16), b3, 0u, 0u);
    }
}

#ifndef LB_ABL
#define LB_ABL 0          // 1: no hit bodies  2: no test at all  3: no MFMAs (LDS reads stay)  4: no LDS reads either  5: no barrier (wrong)
#endif
#ifndef LB_NW
#define LB_NW 8           // waves per block (x UG groups of 32 users each)
#endif
#ifndef LB_PRIV
#define LB_PRIV 0         // 1: a list per (user, lane half) with a register counter -- no LDS atomics
#endif
template <int D, int UG>
struct List2Cfg {
    static constexpr int RSB = 2 * D + 8;                     // row: hi[D], lo[D], ext[8] (in what was padding)
    static constexpr int NS = D / 16;
    static constexpr int NW = LB_NW, THREADS = 64 * NW, UPB = 32 * NW * UG;
    static constexpr int UNITS = kTileItems * 2 * D / 8;
    static constexpr size_t smem = (size_t)2 * kTileItems * RSB * 2 + UPB * 4 + 16;
};

#ifdef LB_TRACE
__device__ unsigned long long g_trace[64][8];
#define LB_T(i) do { if (trace_on) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_tr[(i)] = __builtin_readcyclecounter(); } } while (0)
#else
#define LB_T(i) do { } while (0)
#endif
template <int D, int UG, int WPS>
__global__ __launch_bounds__((List2Cfg<D, UG>::THREADS), WPS) void k_list2(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items_bf, const uint4 *__restrict__ items_ext,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits, const float *__restrict__ sig_u, float c,
    const uint32_t *__restrict__ mask_bits, int item_offset, int ublocks, const float *__restrict__ tau,
    uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow) {
    using C = List2Cfg<D, UG>;
    constexpr int THREADS = C::THREADS, RSB = C::RSB, NS = C::NS, UPB = C::UPB;
    constexpr int LDU = (C::UNITS + THREADS - 1) / THREADS;
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *s_a = reinterpret_cast<__bf16 *>(smem);                                   // [2][32][RSB]
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(smem + (size_t)2 * kTileItems * RSB * 2);   // [256]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    const int n_ub = ublocks;
    const long long G = gridDim.x, b = blockIdx.x;
    int S = (int)(0.6180339f * (float)T);
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
    // the user-side ext slab: ones against the three bias terms (k = 0, 1, 2), zeros elsewhere
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = h == 0 ? 0x3f803f80u : 0u; ones.u[1] = h == 0 ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    const bf16x8 bext = ones.v;
    for (long long w = W * b / G; w < w_end;) {
        const int ub = (int)(w / T), i0 = (int)(w - (long long)ub * T);
        const int i1 = (int)min((long long)T, i0 + (w_end - w));
        w += i1 - i0;
        long long first = (long long)ub * T * G / W;
        while (W * (first + 1) / G <= (long long)ub * T) ++first;
        while (W * first / G > (long long)ub * T) --first;
        const int split = (int)(b - first);
        for (int k = tid; k < UPB; k += THREADS) s_cnt[k] = 0u;
        int uslot[UG], q[UG];
        uint32_t my_cnt[UG];
        bool q_ok[UG];
        bf16x8 bhi[UG][NS], blo[UG][NS];
        float su[UG], tau_s[UG], thr[UG];
        uint64_t *my_list[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            uslot[g] = (wid * UG + g) * 32 + col;
            q[g] = ub * UPB + uslot[g];
            my_cnt[g] = 0u;
            q_ok[g] = q[g] < U;
            const uint4 *urow = users_bf + (size_t)(q_ok[g] ? q[g] : 0) * (2 * D / 8);
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
                if (!q_ok[g]) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
                bhi[g][sI] = *reinterpret_cast<bf16x8 *>(&v);
                blo[g][sI] = *reinterpret_cast<bf16x8 *>(&l);
            }
            su[g] = q_ok[g] ? sig_u[q[g]] : 1.0f;
            tau_s[g] = __builtin_nanf("");
            if (q_ok[g]) tau_s[g] = tau[q[g]] - 1.01f * filter_margin(D, unorm[q[g]], qmax, c);
            thr[g] = tau_s[g] / su[g];                         // listing test on acc'' (RUBI_BOTH)
            my_list[g] = LB_PRIV ? lists + ((size_t)(2 * split + h) * U + (q_ok[g] ? q[g] : 0)) * (cap / 2)
                                 : lists + ((size_t)split * U + (q_ok[g] ? q[g] : 0)) * cap;
        }
        int vi = i0, t = visit(i0);
        uint4 stg[LDU], stx = make_uint4(0u, 0u, 0u, 0u);
        uint32_t tm_next[UG];
        auto load_tile = [&](int tile) {
#pragma unroll
            for (int k = 0; k < LDU; ++k) {
                const int e = tid + THREADS * k, row = (e / (2 * D / 8)) & (kTileItems - 1), c8 = e % (2 * D / 8);
                const int it = min(tile * kTileItems + row, n_local - 1);
                stg[k] = items_bf[(size_t)it * (2 * D / 8) + c8];
            }
            if (tid < kTileItems) stx = items_ext[min(tile * kTileItems + tid, n_local - 1)];
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
            if (tid < kTileItems) *reinterpret_cast<uint4 *>(s_a + ((size_t)buf * kTileItems + tid) * RSB + 2 * D) = stx;
        };
        int buf = 0;
        if (vi < i1) { load_tile(t); store_tile(0); }
        uint32_t tm_cur[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
        __syncthreads();
        while (vi < i1) {
#ifdef LB_TRACE
            unsigned long long g_tr[8];
            const bool trace_on = blockIdx.x == LB_TRACE && wid == 3 && vi - i0 >= 20 && vi - i0 < 84;
#endif
            LB_T(0);
            const bool has_next = vi + 1 < i1;
            const int tn = has_next ? visit_after(t) : t;
            if (has_next) load_tile(tn);
            LB_T(1);
            const int gid0 = t * kTileItems + item_offset;
            const int valid = n_local - t * kTileItems;
            const uint32_t tail = valid < kTileItems ? (valid > 0 ? ~0u << valid : ~0u) : 0u;
            const __bf16 *ua = s_a + ((size_t)buf * kTileItems + col) * RSB + 8 * h;
            f32x16 acc[UG];
#pragma unroll
            for (int g = 0; g < UG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#if LB_ABL < 4
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
                bf16x8 al = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
#if LB_ABL == 3
                asm volatile("" :: "v"(ah), "v"(al));
#else
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[g][sI], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[g][sI], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[g][sI], acc[g], 0, 0, 0);
#endif
            }
            {   // the bias slab, last (both lane halves read the item's 16 bytes; the user side is zero for k >= 8)
                bf16x8 ae = *reinterpret_cast<const bf16x8 *>(s_a + ((size_t)buf * kTileItems + col) * RSB + 2 * D);
#if LB_ABL == 3
                asm volatile("" :: "v"(ae));
#else
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc[g], 0, 0, 0);
#endif
            }
#endif
#ifdef LB_TRACE
            if (trace_on) { asm volatile("s_nop 0" :: "v"(acc[0][0])); }
#endif
            LB_T(2);
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                uint32_t hit = 0u;
#if LB_ABL >= 2
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[g][r]));
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) hit |= __ballot(acc[g][r] >= thr[g]) ? 1u << r : 0u;
#endif
#if LB_ABL == 1
                if (hit == 0x12345u) s_cnt[0] = 1u;
                hit = 0u;
#endif
                if (hit) {
                    const uint32_t tmask = tm_cur[g] | tail;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (!((hit >> r) & 1u)) continue;
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                        bool pass = acc[g][r] >= thr[g];
                        pass = pass && ((tmask >> row) & 1u) == 0u;
                        if (pass) {
                            const float v = acc[g][r] * su[g];
                            if (v >= tau_s[g]) {
#if LB_PRIV
                                if (my_cnt[g] < (uint32_t)(cap / 2)) my_list[g][my_cnt[g]++] = make_key(v, gid0 + row);
                                else { overflow[0] = 1; tau_s[g] = INFINITY; thr[g] = INFINITY; }
#else
                                const uint32_t pos = atomicAdd(&s_cnt[uslot[g]], 1u);
                                if (pos < (uint32_t)cap) my_list[g][pos] = make_key(v, gid0 + row);
                                else { overflow[0] = 1; tau_s[g] = INFINITY; thr[g] = INFINITY; }
#endif
                            }
                        }
                    }
                }
            }
            LB_T(3);
#ifdef LB_TRACE
            if (trace_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
            LB_T(4);
            if (has_next) store_tile(buf ^ 1);
            LB_T(5);
#if LB_ABL != 5
            __syncthreads();
#endif
            LB_T(6);
#ifdef LB_TRACE
            if (trace_on && lane == 0) for (int k = 0; k < 7; ++k) g_trace[vi - i0 - 20][k] = g_tr[k];
#endif
#pragma unroll
            for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
            buf ^= 1;
            t = tn; ++vi;
        }
#if LB_PRIV
#pragma unroll
        for (int g = 0; g < UG; ++g)
            if (q_ok[g]) counts[(size_t)(2 * split + h) * U + q[g]] = (int32_t)my_cnt[g];
#else
        for (int k = tid; k < UPB; k += THREADS) {
            const int qq = ub * UPB + k;
            if (qq < U) counts[(size_t)split * U + qq] = (int32_t)min(s_cnt[k], (uint32_t)cap);
        }
#endif
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Rewrite 2: the tile arrives by LDS-DMA two visits ahead.
//   item rows in global memory as they sit in LDS: 17 units of 16 bytes (hi[D], lo[D], ext[8]: 272 B -- the odd unit stride
//   that makes the fragment reads conflict-free), a tile = 544 consecutive units, copied verbatim by nine
//   global_load_lds_dwordx4 (no staging registers, no ds_write).  Three LDS buffers; the four even waves fetch the even
//   visits, the four odd waves the odd ones, so a wave's `s_waitcnt vmcnt(0)` (its list stores share the counter) waits for
//   a copy it issued TWO visits ago.  One barrier per visit.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRowUnits3 = 17, kTileUnits3 = kTileItems * kRowUnits3, kBufUnits3 = 576;      // (9 x 64: the ninth copy is half padding)

template <int D>
__global__ __launch_bounds__(256) void k_prep_items3(int n, int n_pad, const float *__restrict__ items, const float *__restrict__ sig_i, float c,
                                                     uint4 *__restrict__ items3) {
    constexpr int LPRB = D / 8;
    static_assert(2 * LPRB + 1 == kRowUnits3, "d = 64");
    const int sub = threadIdx.x % LPRB, row = blockIdx.x * (256 / LPRB) + threadIdx.x / LPRB;
    if (row >= n_pad) return;
    uint4 *dst = items3 + (size_t)row * kRowUnits3;
    if (row >= n) {                                            // rows past the end: zeros (never listed: tail mask)
        dst[sub] = make_uint4(0u, 0u, 0u, 0u); dst[LPRB + sub] = make_uint4(0u, 0u, 0u, 0u);
        if (sub == 0) dst[2 * LPRB] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const float s = sig_i[row];
    float4 a = ld4(items + (size_t)row * D + 8 * sub), b = ld4(items + (size_t)row * D + 8 * sub + 4);
    const float x[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        hi[k] = bf16_rne_bits(x[k]);
        lo[k] = bf16_rne_bits(x[k] - __uint_as_float(hi[k] << 16));
    }
    dst[sub] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    dst[LPRB + sub] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    if (sub == 0) {
        const float bias = -c * s;
        const uint32_t b1 = bf16_rne_bits(bias);
        const float r1 = bias - __uint_as_float(b1 << 16);
        const uint32_t b2 = bf16_rne_bits(r1);
        const uint32_t b3 = bf16_rne_bits(r1 - __uint_as_float(b2 << 16));
        dst[2 * LPRB] = make_uint4(b1 | (b2 << 16), b3, 0u, 0u);
    }
}

#ifndef LB_DIST
#define LB_DIST 2
#endif
template <int D, int WPS>
__global__ __launch_bounds__(512, WPS) void k_list3(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items3,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits, const float *__restrict__ sig_u, float c,
    const uint32_t *__restrict__ mask_bits, int item_offset, int ublocks, const float *__restrict__ tau,
    uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow) {
    constexpr int THREADS = 512, NS = D / 16, RSB = 2 * D + 8;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    __shared__ uint4 s_t[3][kBufUnits3];
    __shared__ uint32_t s_cnt[kUsersPerBlock];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    const long long G = gridDim.x, b = blockIdx.x;
    int S = (int)(0.6180339f * (float)T);
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
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = h == 0 ? 0x3f803f80u : 0u; ones.u[1] = h == 0 ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    const bf16x8 bext = ones.v;
    const int my_par = wid & 1, my_q = wid >> 1;               // this wave copies visits of parity my_par: units my_q*128 .. +128 (+ the ext units: my_q == 0)
    auto copy_tile = [&](int tile, int bufn) {                 // called by the waves of the visit's parity only
        const uint4 *src = items3 + (size_t)tile * kTileUnits3 + my_q * 128 + lane;
        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)&s_t[bufn][my_q * 128], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(src + 64), (lds_void *)&s_t[bufn][my_q * 128 + 64], 16, 0, 0);
        if (my_q == 0)
            __builtin_amdgcn_global_load_lds((glb_void *)(items3 + (size_t)tile * kTileUnits3 + 512 + lane), (lds_void *)&s_t[bufn][512], 16, 0, 0);
    };
    for (long long w = W * b / G; w < w_end;) {
        const int ub = (int)(w / T), i0 = (int)(w - (long long)ub * T);
        const int i1 = (int)min((long long)T, i0 + (w_end - w));
        w += i1 - i0;
        long long first = (long long)ub * T * G / W;
        while (W * (first + 1) / G <= (long long)ub * T) ++first;
        while (W * first / G > (long long)ub * T) --first;
        const int split = (int)(b - first);
        __syncthreads();                                       // (the previous segment's readers are done with s_t and s_cnt)
        for (int k = tid; k < kUsersPerBlock; k += THREADS) s_cnt[k] = 0u;
        const int uslot = wid * 32 + col, q = ub * kUsersPerBlock + uslot;
        const bool q_ok = q < U;
        bf16x8 bhi[NS], blo[NS];
        const uint4 *urow = users_bf + (size_t)(q_ok ? q : 0) * (2 * D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
            if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[sI] = *reinterpret_cast<bf16x8 *>(&v);
            blo[sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
        const float su = q_ok ? sig_u[q] : 1.0f;
        float tau_s = __builtin_nanf("");
        if (q_ok) tau_s = tau[q] - 1.01f * filter_margin(D, unorm[q], qmax, c);
        float thr = tau_s / su;
        uint64_t *my_list = lists + ((size_t)split * U + (q_ok ? q : 0)) * cap;
        // prologue: visits i0 (and i0 + 1) on their way; visit v lives in buffer (v - i0) % 3
        int t = visit(i0), t1 = visit_after(t);
        if (my_par == 0) copy_tile(t, 0);
        if (my_par == 1 && i0 + 1 < i1) copy_tile(t1, 1);
        uint32_t tm_cur = (mask_bits && q_ok) ? mask_bits[(size_t)t * U + q] : 0u;
        int bufi = 0;
        for (int vi = i0; vi < i1; ++vi) {
            const int par = (vi - i0) & 1;
            if (par == my_par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my copy of THIS visit (issued two visits ago)
            __syncthreads();
            const int tn = visit_after(t), tn2 = visit_after(tn);
            if (LB_DIST == 2 && par == my_par && vi + 2 < i1) copy_tile(tn2, bufi == 0 ? 2 : bufi - 1);
            const uint32_t tm_next = (mask_bits && q_ok && vi + 1 < i1) ? mask_bits[(size_t)tn * U + q] : 0u;
            const int gid0 = t * kTileItems + item_offset;
            const int valid = n_local - t * kTileItems;
            const uint32_t tail = valid < kTileItems ? (valid > 0 ? ~0u << valid : ~0u) : 0u;
            const __bf16 *ua = reinterpret_cast<const __bf16 *>(&s_t[bufi][0]) + (size_t)col * RSB + 8 * h;
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
            {
                const bf16x8 ae = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(&s_t[bufi][0]) + (size_t)col * RSB + 2 * D);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc, 0, 0, 0);
            }
            uint32_t hit = 0u;
#if LB_ABL >= 2
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[r]));
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) hit |= __ballot(acc[r] >= thr) ? 1u << r : 0u;
#endif
            if (hit) {
                const uint32_t tmask = tm_cur | tail;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!((hit >> r) & 1u)) continue;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    bool pass = acc[r] >= thr;
                    pass = pass && ((tmask >> row) & 1u) == 0u;
                    if (pass) {
                        const float v = acc[r] * su;
                        if (v >= tau_s) {
                            const uint32_t pos = atomicAdd(&s_cnt[uslot], 1u);
                            if (pos < (uint32_t)cap) my_list[pos] = make_key(v, gid0 + row);
                            else { overflow[0] = 1; tau_s = INFINITY; thr = INFINITY; }
                        }
                    }
                }
            }
            if (LB_DIST == 1 && vi + 1 < i1) {                 // (A/B: the copy one visit ahead, as the product kernel fetches)
                if (par != my_par) copy_tile(tn, bufi == 2 ? 0 : bufi + 1);
            }
            tm_cur = tm_next;
            bufi = bufi == 2 ? 0 : bufi + 1;
            t = tn;
        }
        __syncthreads();
        for (int k = tid; k < kUsersPerBlock; k += THREADS) {
            const int qq = ub * kUsersPerBlock + k;
            if (qq < U) counts[(size_t)split * U + qq] = (int32_t)min(s_cnt[k], (uint32_t)cap);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Rewrite 3: lean on the vector ALU.  The trace (LB_TRACE) says a visit costs a wave ~2500 cycles of which the MFMAs are a
// third: the rest is VALU work that four waves per SIMD queue up for -- address arithmetic of the tile copy, the 16
// compares, the bodies of the hits.  Here: item rows of 17 units in global memory as in LDS (tile = 544 consecutive units:
// thread t copies unit t, wave-uniform tile offsets stay in SGPRs), the visit loop unrolled over the two LDS buffers
// (immediate offsets), one compare + one branch per score, a hit body of eight vector instructions.
// ---------------------------------------------------------------------------------------------------------------------
__device__ unsigned long long g_blk[1024][4];       // per block: start, end (s_memtime), hardware id, first visit's start
template <int D, int WPS>
__global__ __launch_bounds__(512, WPS) void k_list4(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items3,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits, const float *__restrict__ sig_u, float c,
    const uint32_t *__restrict__ mask_bits, int item_offset, int ublocks, const float *__restrict__ tau,
    uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow) {
    constexpr int THREADS = 512, NS = D / 16, RU = 2 * D / 8 + 1, TU = kTileItems * RU;     // units per row / per tile
    constexpr int LDU = (TU + THREADS - 1) / THREADS;
    __shared__ uint4 s_t[2][TU];
    __shared__ uint32_t s_cnt[kUsersPerBlock];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    const long long G = gridDim.x, b = blockIdx.x;
    int S = (int)(0.6180339f * (float)T);
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
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = h == 0 ? 0x3f803f80u : 0u; ones.u[1] = h == 0 ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    const bf16x8 bext = ones.v;
    const uint4 *my_src = items3 + tid;                       // + tile * TU (wave-uniform)
#ifdef LB_BLKTIME
    if (tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_blk[blockIdx.x][0] = __builtin_readcyclecounter(); g_blk[blockIdx.x][2] = ((unsigned long long)xcc << 32) | hw;
        g_blk[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#ifndef LB_PRIO
#define LB_PRIO 0         // two blocks share a CU and fall into step (the matrix pipe serves their four waves evenly, so they leave the
#endif                    // MFMA phase together and idle together): 1..3 = ways of giving one of them priority
#if LB_PRIO == 1
    if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(3);
#elif LB_PRIO == 2
    if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(3);
#elif LB_PRIO == 3
    if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_setprio(3);
#elif LB_PRIO == 4
    if ((blockIdx.x / 256) & 1) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
#endif
    for (long long w = W * b / G; w < w_end;) {
        const int ub = (int)(w / T), i0 = (int)(w - (long long)ub * T);
        const int i1 = (int)min((long long)T, i0 + (w_end - w));
        w += i1 - i0;
        long long first = (long long)ub * T * G / W;
        while (W * (first + 1) / G <= (long long)ub * T) ++first;
        while (W * first / G > (long long)ub * T) --first;
        const int split = (int)(b - first);
        __syncthreads();
        for (int k = tid; k < kUsersPerBlock; k += THREADS) s_cnt[k] = 0u;
        const int uslot = wid * 32 + col, q = ub * kUsersPerBlock + uslot;
        const bool q_ok = q < U;
        bf16x8 bhi[NS], blo[NS];
        const uint4 *urow = users_bf + (size_t)(q_ok ? q : 0) * (2 * D / 8);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
            uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
            if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
            bhi[sI] = *reinterpret_cast<bf16x8 *>(&v);
            blo[sI] = *reinterpret_cast<bf16x8 *>(&l);
        }
        const float su = q_ok ? sig_u[q] : 1.0f;
        float tau_s = __builtin_nanf("");
        if (q_ok) tau_s = tau[q] - 1.01f * filter_margin(D, unorm[q], qmax, c);
        float thr = tau_s / su;                                // NaN for padding queries: never listed
#if LB_PRIV
        uint64_t *my_list = lists + ((size_t)(2 * split + h) * U + (q_ok ? q : 0)) * (cap / 2);
        uint32_t my_n = 0u;
#else
        uint64_t *my_list = lists + ((size_t)split * U + (q_ok ? q : 0)) * cap;
#endif
        const uint32_t *my_mask = mask_bits + (q_ok ? q : 0);  // + tile * U (wave-uniform)
        uint32_t *my_cnt = &s_cnt[uslot];
        const int id_lane = item_offset + 4 * h;               // id of accumulator slot r: tile * 32 + (r & 3) + 8 * (r >> 2) + id_lane

        uint4 stg[LDU];
        uint32_t tm_next = 0u;
        // every load unconditional (a load under a branch makes the compiler wait for it at the join): the last, partial
        // round of units is fetched by everybody from a wrapped address and stored by the threads it belongs to
        constexpr int REM = TU - THREADS * (LDU - 1);          // units of the last round (1 .. THREADS)
        const int last_off = THREADS * (LDU - 1) + (tid < REM ? 0 : tid % REM - tid);
        auto load_tile = [&](int tile) {
            const uint4 *src = my_src + (size_t)__builtin_amdgcn_readfirstlane(tile) * TU;
#pragma unroll
            for (int k = 0; k + 1 < LDU; ++k) stg[k] = src[THREADS * k];
            stg[LDU - 1] = src[last_off];
            tm_next = my_mask[(size_t)__builtin_amdgcn_readfirstlane(tile) * U];
        };
        auto store_tile = [&](auto BUF) {
            constexpr int buf = decltype(BUF)::value;
#pragma unroll
            for (int k = 0; k < LDU; ++k) {
                asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
                if (k + 1 < LDU || tid < REM) s_t[buf][tid + THREADS * k] = stg[k];
            }
        };
        unsigned long long g_tr2 = 0;
        // one visit: tile t sits in buffer BUF; the next tile's copy is in flight in stg
        auto one_visit = [&](auto BUF, int t, uint32_t tm_cur) {
            constexpr int buf = decltype(BUF)::value;
            const __bf16 *ua = reinterpret_cast<const __bf16 *>(&s_t[buf][0]) + (size_t)col * (8 * RU) + 8 * h;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#ifndef LB_CHAIN
#define LB_CHAIN 1        // 1: every fragment in registers first, then the MFMAs back to back (an instruction between two MFMAs on one accumulator costs ~43 cycles)
#endif
#if LB_CHAIN
            bf16x8 ah[NS], al[NS];
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                ah[sI] = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
                al[sI] = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
            }
            const bf16x8 ae = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(&s_t[buf][0]) + (size_t)col * (8 * RU) + 2 * D);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[sI], bhi[sI], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sI], blo[sI], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sI], bhi[sI], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef LB_TRACE
            if (blockIdx.x == LB_TRACE && wid == 3) { asm volatile("s_nop 0" :: "v"(acc[0])); g_tr2 = __builtin_readcyclecounter(); }
#endif
#else
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[sI], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[sI], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[sI], acc, 0, 0, 0);
            }
            {
                const bf16x8 ae = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(&s_t[buf][0]) + (size_t)col * (8 * RU) + 2 * D);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc, 0, 0, 0);
            }
#endif
            const int valid = n_local - t * kTileItems;       // < 32 only in the last tile of the shard
            uint32_t tmask = tm_cur;
            if (valid < kTileItems) tmask |= ~0u << (valid > 0 ? valid : 0);
            const uint32_t tmh = tmask >> (4 * h);            // bit (r & 3) + 8 * (r >> 2): this lane's row of slot r is masked
            const int id0 = t * kTileItems + id_lane;
#if LB_ABL >= 2
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[r]));
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool above = acc[r] >= thr;
                if (__builtin_amdgcn_ballot_w64(above)) {      // wave-uniform: some lane's score r passes
                    const int rbit = (r & 3) + 8 * (r >> 2);
                    uint32_t m = tmh;
                    asm volatile("" : "+v"(m));                // (the mask test belongs in here, not in front of the branch)
                    if (above && !((m >> rbit) & 1u)) {
#if LB_PRIV
                        if (my_n < (uint32_t)(cap / 2)) my_list[my_n] = make_key(acc[r] * su, id0 + rbit);
                        my_n += 1u;
#else
                        const uint32_t pos = atomicAdd(my_cnt, 1u);        // (keeps counting past cap: the segment's end flags it)
                        if (pos < (uint32_t)cap) my_list[pos] = make_key(acc[r] * su, id0 + rbit);
#endif
                    }
                }
            }
#endif
        };
        int t = visit(i0);
        if (i0 < i1) { load_tile(t); store_tile(std::integral_constant<int, 0>()); }
        uint32_t tm_cur = tm_next;
        __syncthreads();
        int vi = i0;
        while (vi < i1) {
            {
#ifdef LB_TRACE
                unsigned long long g_tr[8];
                const bool trace_on = blockIdx.x == LB_TRACE && wid == 3 && vi - i0 >= 20 && vi - i0 < 148;
#endif
                LB_T(0);
                const bool has_next = vi + 1 < i1;
                const int tn = has_next ? visit_after(t) : t;
                load_tile(tn);
                __builtin_amdgcn_sched_barrier(0);
                LB_T(1);
                one_visit(std::integral_constant<int, 0>(), t, tm_cur);
                LB_T(3);
#ifdef LB_TRACE
                if (trace_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
                LB_T(4);
                store_tile(std::integral_constant<int, 1>());
                LB_T(5);
                __syncthreads();
                LB_T(6);
#ifdef LB_TRACE
                if (trace_on && lane == 0) { g_tr[2] = g_tr2; for (int k = 0; k < 7; ++k) g_trace[(vi - i0 - 20) / 2][k] = g_tr[k]; }
#endif
                tm_cur = tm_next; t = tn; ++vi;
            }
            if (vi >= i1) break;
            {
                const bool has_next = vi + 1 < i1;
                const int tn = has_next ? visit_after(t) : t;
                load_tile(tn);
                __builtin_amdgcn_sched_barrier(0);
                one_visit(std::integral_constant<int, 1>(), t, tm_cur);
                store_tile(std::integral_constant<int, 0>());
                __syncthreads();
                tm_cur = tm_next; t = tn; ++vi;
            }
        }
#if LB_PRIV
        if (q_ok) {
            counts[(size_t)(2 * split + h) * U + q] = (int32_t)min(my_n, (uint32_t)(cap / 2));
            if (my_n > (uint32_t)(cap / 2)) overflow[0] = 1;
        }
#else
        for (int k = tid; k < kUsersPerBlock; k += THREADS) {
            const int qq = ub * kUsersPerBlock + k;
            if (qq < U) {
                counts[(size_t)split * U + qq] = (int32_t)min(s_cnt[k], (uint32_t)cap);
                if (s_cnt[k] > (uint32_t)cap) overflow[0] = 1;
            }
        }
#endif
    }
#ifdef LB_BLKTIME
    if (tid == 0) g_blk[blockIdx.x][1] = __builtin_readcyclecounter();
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Rewrite 4: k_list4 with UG groups of 32 queries per wave (every item fragment read from LDS feeds UG x 13 MFMAs, one
// barrier per UG x 13 MFMAs) in blocks of NW waves.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int UG, int NW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void k_list5(
    int U, int n_local, const uint4 *__restrict__ users_bf, const uint4 *__restrict__ items3,
    const float *__restrict__ unorm, const uint32_t *__restrict__ qmax_bits, const float *__restrict__ sig_u, float c,
    const uint32_t *__restrict__ mask_bits, int item_offset, int ublocks, const float *__restrict__ tau,
    uint64_t *__restrict__ lists, int32_t *__restrict__ counts, int cap, int32_t *overflow) {
    constexpr int THREADS = 64 * NW, UPB = 32 * NW * UG, NS = D / 16, RU = 2 * D / 8 + 1, TU = kTileItems * RU;
    constexpr int LDU = (TU + THREADS - 1) / THREADS, REM = TU - THREADS * (LDU - 1);
    __shared__ uint4 s_t[2][TU];
    __shared__ uint32_t s_cnt[UPB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int T = (n_local + kTileItems - 1) / kTileItems;
    const long long G = gridDim.x, b = blockIdx.x;
    int S = (int)(0.6180339f * (float)T);
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
    union { uint32_t u[4]; bf16x8 v; } ones;
    ones.u[0] = h == 0 ? 0x3f803f80u : 0u; ones.u[1] = h == 0 ? 0x00003f80u : 0u; ones.u[2] = 0u; ones.u[3] = 0u;
    const bf16x8 bext = ones.v;
    const uint4 *my_src = items3 + tid;
    const int last_off = THREADS * (LDU - 1) + (tid < REM ? 0 : tid % REM - tid);
    for (long long w = W * b / G; w < w_end;) {
        const int ub = (int)(w / T), i0 = (int)(w - (long long)ub * T);
        const int i1 = (int)min((long long)T, i0 + (w_end - w));
        w += i1 - i0;
        long long first = (long long)ub * T * G / W;
        while (W * (first + 1) / G <= (long long)ub * T) ++first;
        while (W * first / G > (long long)ub * T) --first;
        const int split = (int)(b - first);
        __syncthreads();
        for (int k = tid; k < UPB; k += THREADS) s_cnt[k] = 0u;
        bf16x8 bhi[UG][NS], blo[UG][NS];
        float su[UG], thr[UG];
        uint64_t *my_list[UG];
        const uint32_t *my_mask[UG];
        uint32_t *my_cnt[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            const int uslot = (wid * UG + g) * 32 + col, q = ub * UPB + uslot;
            const bool q_ok = q < U;
            const uint4 *urow = users_bf + (size_t)(q_ok ? q : 0) * (2 * D / 8);
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                uint4 v = urow[2 * sI + h], l = urow[D / 8 + 2 * sI + h];
                if (!q_ok) { v = make_uint4(0u, 0u, 0u, 0u); l = v; }
                bhi[g][sI] = *reinterpret_cast<bf16x8 *>(&v);
                blo[g][sI] = *reinterpret_cast<bf16x8 *>(&l);
            }
            su[g] = q_ok ? sig_u[q] : 1.0f;
            float tau_s = __builtin_nanf("");
            if (q_ok) tau_s = tau[q] - 1.01f * filter_margin(D, unorm[q], qmax, c);
            thr[g] = tau_s / su[g];
            my_list[g] = lists + ((size_t)split * U + (q_ok ? q : 0)) * cap;
            my_mask[g] = mask_bits + (q_ok ? q : 0);
            my_cnt[g] = &s_cnt[uslot];
        }
        const int id_lane = item_offset + 4 * h;
        uint4 stg[LDU];
        uint32_t tm_next[UG];
        auto load_tile = [&](int tile) {
            const uint4 *src = my_src + (size_t)__builtin_amdgcn_readfirstlane(tile) * TU;
#pragma unroll
            for (int k = 0; k + 1 < LDU; ++k) stg[k] = src[THREADS * k];
            stg[LDU - 1] = src[last_off];
#pragma unroll
            for (int g = 0; g < UG; ++g) tm_next[g] = my_mask[g][(size_t)__builtin_amdgcn_readfirstlane(tile) * U];
        };
        auto store_tile = [&](auto BUF) {
            constexpr int buf = decltype(BUF)::value;
#pragma unroll
            for (int k = 0; k < LDU; ++k) {
                asm volatile("" : "+v"(stg[k].x), "+v"(stg[k].y), "+v"(stg[k].z), "+v"(stg[k].w));
                if (k + 1 < LDU || tid < REM) s_t[buf][tid + THREADS * k] = stg[k];
            }
        };
        auto one_visit = [&](auto BUF, int t, const uint32_t (&tm_cur)[UG]) {
            constexpr int buf = decltype(BUF)::value;
            const __bf16 *ua = reinterpret_cast<const __bf16 *>(&s_t[buf][0]) + (size_t)col * (8 * RU) + 8 * h;
            f32x16 acc[UG];
#pragma unroll
            for (int g = 0; g < UG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
            bf16x8 ah[NS], al[NS];
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
                ah[sI] = *reinterpret_cast<const bf16x8 *>(ua + 16 * sI);
                al[sI] = *reinterpret_cast<const bf16x8 *>(ua + D + 16 * sI);
            }
            const bf16x8 ae = *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const __bf16 *>(&s_t[buf][0]) + (size_t)col * (8 * RU) + 2 * D);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sI = 0; sI < NS; ++sI) {
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[sI], bhi[g][sI], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sI], blo[g][sI], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sI], bhi[g][sI], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < UG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ae, bext, acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int valid = n_local - t * kTileItems;
            const uint32_t tailm = valid < kTileItems ? ~0u << (valid > 0 ? valid : 0) : 0u;
            const int id0 = t * kTileItems + id_lane;
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                const uint32_t tmh = (tm_cur[g] | tailm) >> (4 * h);
#if LB_ABL >= 2
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(acc[g][r]));
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool above = acc[g][r] >= thr[g];
                    if (__builtin_amdgcn_ballot_w64(above)) {
                        const int rbit = (r & 3) + 8 * (r >> 2);
                        uint32_t m = tmh;
                        asm volatile("" : "+v"(m));
                        if (above && !((m >> rbit) & 1u)) {
                            const uint32_t pos = atomicAdd(my_cnt[g], 1u);
                            if (pos < (uint32_t)cap) my_list[g][pos] = make_key(acc[g][r] * su[g], id0 + rbit);
                        }
                    }
                }
#endif
            }
        };
        int t = visit(i0);
        if (i0 < i1) { load_tile(t); store_tile(std::integral_constant<int, 0>()); }
        uint32_t tm_cur[UG];
#pragma unroll
        for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
        __syncthreads();
        int vi = i0;
        while (vi < i1) {
            {
                const int tn = vi + 1 < i1 ? visit_after(t) : t;
                load_tile(tn);
                __builtin_amdgcn_sched_barrier(0);
                one_visit(std::integral_constant<int, 0>(), t, tm_cur);
                store_tile(std::integral_constant<int, 1>());
                __syncthreads();
#pragma unroll
                for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
                t = tn; ++vi;
            }
            if (vi >= i1) break;
            {
                const int tn = vi + 1 < i1 ? visit_after(t) : t;
                load_tile(tn);
                __builtin_amdgcn_sched_barrier(0);
                one_visit(std::integral_constant<int, 1>(), t, tm_cur);
                store_tile(std::integral_constant<int, 0>());
                __syncthreads();
#pragma unroll
                for (int g = 0; g < UG; ++g) tm_cur[g] = tm_next[g];
                t = tn; ++vi;
            }
        }
        for (int k = tid; k < UPB; k += THREADS) {
            const int qq = ub * UPB + k;
            if (qq < U) {
                counts[(size_t)split * U + qq] = (int32_t)min(s_cnt[k], (uint32_t)cap);
                if (s_cnt[k] > (uint32_t)cap) overflow[0] = 1;
            }
        }
    }
}

// exact fp32 scores of `ns` sampled items per user (threshold calibration of the bench)
__global__ void k_sample_scores(int U, int d, int ns, const float *__restrict__ P, const float *__restrict__ Q, const int32_t *__restrict__ pick,
                                const float *__restrict__ su, const float *__restrict__ si, float c, float *__restrict__ out) {
    const int u = blockIdx.x, j = blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= ns) return;
    const int it = pick[j];
    float a = 0.f;
    for (int k = 0; k < d; ++k) a = fmaf(P[(size_t)u * d + k], Q[(size_t)it * d + k], a);
    out[(size_t)u * ns + j] = ((a - c) * si[it]) * su[u];
}

int main(int argc, char **argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 15424, N = argc > 2 ? atoi(argv[2]) : 40981;
    const int cand = argc > 3 ? atoi(argv[3]) : 170, reps = argc > 4 ? atoi(argv[4]) : 20;
    constexpr int D = 64, KIND = MACR_SCORE_RUBI_BOTH;
    const float c = 40.f;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> P((size_t)U * D), Q((size_t)N * D), su(U), si(N);
    // a trained model's shape: item rows whose length falls with the id (popular items first), users of similar length
    for (int i = 0; i < N; ++i) {
        const float len = 0.35f * (0.3f + 0.7f / (1.f + 4.f * (float)i / N));
        for (int k = 0; k < D; ++k) Q[(size_t)i * D + k] = nd(rng) * len;
        si[i] = 1.f / (1.f + expf(-(nd(rng) * 0.8f - 0.5f - 1.5f * (float)i / N)));
    }
    for (int u = 0; u < U; ++u) {
        for (int k = 0; k < D; ++k) P[(size_t)u * D + k] = nd(rng) * 0.3f;
        su[u] = 1.f / (1.f + expf(-(nd(rng) * 0.5f)));
    }
    float *dP, *dQ, *dsu, *dsi;
    CK(hipMalloc(&dP, P.size() * 4)); CK(hipMalloc(&dQ, Q.size() * 4)); CK(hipMalloc(&dsu, U * 4)); CK(hipMalloc(&dsi, N * 4));
    CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsu, su.data(), U * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsi, si.data(), N * 4, hipMemcpyHostToDevice));

    // thresholds: the score at the rank that leaves ~cand items above it, estimated on 4096 sampled items per user
    const int ns = 4096;
    std::vector<int32_t> pick(ns);
    for (int j = 0; j < ns; ++j) pick[j] = (int)(rng() % (unsigned)N);
    int32_t *dpick; float *dsamp, *dtau;
    CK(hipMalloc(&dpick, ns * 4)); CK(hipMalloc(&dsamp, (size_t)U * ns * 4)); CK(hipMalloc(&dtau, U * 4));
    CK(hipMemcpy(dpick, pick.data(), ns * 4, hipMemcpyHostToDevice));
    k_sample_scores<<<dim3(U, ns / 256), 256>>>(U, D, ns, dP, dQ, dpick, dsu, dsi, c, dsamp);
    CK(hipDeviceSynchronize());
    std::vector<float> samp((size_t)U * ns), tau(U);
    CK(hipMemcpy(samp.data(), dsamp, samp.size() * 4, hipMemcpyDeviceToHost));
    const int kth = std::max(1, (int)((double)ns * cand / N));
    for (int u = 0; u < U; ++u) {
        float *r = samp.data() + (size_t)u * ns;
        std::nth_element(r, r + kth - 1, r + ns, [](float a, float b) { return a > b; });
        tau[u] = r[kth - 1];
    }
    CK(hipMemcpy(dtau, tau.data(), U * 4, hipMemcpyHostToDevice));
    CK(hipFree(dsamp));

    const StreamGeo geo = stream_geo(U, N, D);
    TopkWs sz = carve_topk_ws(nullptr, U, N, geo, D);
    void *wsb; CK(hipMalloc(&wsb, sz.bytes)); CK(hipMemset(wsb, 0, sz.bytes));
    TopkWs ws = carve_topk_ws(wsb, U, N, geo, D);
    uint32_t *qmax_bits = reinterpret_cast<uint32_t *>(ws.overflow + 8);
    k_bf16_prep<D><<<bf16_prep_blocks(U, N, D), 256>>>(U, N, dP, nullptr, dQ, ws.users_bf, ws.items_bf, ws.unorm, qmax_bits);
    uint4 *items2, *ext2;
    CK(hipMalloc(&items2, (size_t)N * D * 4)); CK(hipMalloc(&ext2, (size_t)N * 16));
    k_prep_items2<D><<<(N + 31) / 32, 256>>>(N, dQ, dsi, c, items2, ext2);
    CK(hipDeviceSynchronize());
    printf("U=%d N=%d d=%d  grid1=%d slots1=%d cap=%d  ~%d candidates per user\n", U, N, D, geo.grid1, geo.slots1, ws.cap, cand);

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int32_t *counts2; CK(hipMalloc(&counts2, (size_t)4 * geo.slots1 * U * 4));
    uint64_t *lists2; CK(hipMalloc(&lists2, (size_t)2 * geo.slots1 * U * ws.cap * 8));
    ws.counts = counts2; ws.lists = lists2;
    auto total_listed = [&]() {
        std::vector<int32_t> cn((size_t)4 * geo.slots1 * U);
        CK(hipMemcpy(cn.data(), ws.counts, cn.size() * 4, hipMemcpyDeviceToHost));
        long long s = 0; for (int32_t v : cn) s += v;
        int32_t ov = 0; CK(hipMemcpy(&ov, ws.overflow, 4, hipMemcpyDeviceToHost));
        return std::make_pair(s, ov);
    };
    auto run = [&](const char *name, auto launch) {
        CK(hipMemset(ws.counts, 0, (size_t)4 * geo.slots1 * U * 4)); CK(hipMemset(ws.overflow, 0, 4));
        for (int r = 0; r < 3; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        hipError_t err = hipGetLastError();
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        auto tl = total_listed();
        printf("%-44s %8.1f us   listed %lld (%.1f per user)%s%s\n", name, 1e3 * ms / reps, tl.first, (double)tl.first / U,
               tl.second ? "  OVERFLOW" : "", err != hipSuccess ? hipGetErrorString(err) : "");
    };
    {
        auto kern = k_score_stream_b<D, KIND>;
        const size_t smem = StreamCfgB<D>::smem;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        run("product k_score_stream_b", [&]() {
            kern<<<geo.grid1, StreamGroupsB<D>::THREADS, smem>>>(U, N, ws.users_bf, ws.items_bf, ws.unorm, qmax_bits, dsu, dsi, c, nullptr, nullptr, 0,
                                                                   geo.ublocks, dtau, ws.lists, ws.counts, ws.cap, ws.overflow, 0, nullptr, nullptr, nullptr, 0);
        });
    }
#ifndef LB_UG
#define LB_UG 1
#endif
#ifndef LB_WPS
#define LB_WPS 4
#endif
    {
        auto kern = k_list2<D, LB_UG, LB_WPS>;
#ifndef LB_PAD_LDS
#define LB_PAD_LDS 0      // extra dynamic LDS per block: 90000 leaves room for ONE block per CU
#endif
        const size_t smem = List2Cfg<D, LB_UG>::smem + LB_PAD_LDS;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        char name[128]; snprintf(name, sizeof name, "new: UG=%d NW=%d %d waves/SIMD priv=%d abl=%d pad=%d", LB_UG, LB_NW, LB_WPS, LB_PRIV, LB_ABL, LB_PAD_LDS);
        const int upb = List2Cfg<D, LB_UG>::UPB, ublocks2 = (U + upb - 1) / upb;
        const int grid = (int)std::min<long long>((long long)geo.grid1 * 8 / LB_NW, (long long)ublocks2 * ((N + 31) / 32) / 8);
        run(name, [&]() {
            kern<<<grid, List2Cfg<D, LB_UG>::THREADS, smem>>>(U, N, ws.users_bf, items2, ext2, ws.unorm, qmax_bits, dsu, c, nullptr, 0, ublocks2, dtau,
                                                               ws.lists, ws.counts, ws.cap, ws.overflow);
        });
    }
#ifdef LB_TRACE
    auto print_trace = [&]() {
        unsigned long long tr[64][8];
        CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr)));
        printf("visit:  top->loads issued | -> MFMAs done | -> test+hits done | -> next tile arrived | -> LDS store | -> barrier   (cycles; next top)\n");
        double sum[6] = {0};
        for (int i = 0; i < 64; ++i) {
            for (int k = 0; k < 6; ++k) sum[k] += (double)(tr[i][k + 1] - tr[i][k]);
            if (i < 12) {
                printf("  %2d:", i);
                for (int k = 0; k < 6; ++k) printf(" %6llu", tr[i][k + 1] - tr[i][k]);
                if (i + 1 < 64) printf("   | %6llu", tr[i + 1][0] - tr[i][6]);
                printf("\n");
            }
        }
        printf(" avg:"); for (int k = 0; k < 6; ++k) printf(" %6.0f", sum[k] / 64); printf("   total %.0f per visit (%.0f between tops)\n", (sum[0]+sum[1]+sum[2]+sum[3]+sum[4]+sum[5]) / 64, (double)(tr[63][0] - tr[0][0]) / 63);
    };
    print_trace();
#endif
    {
        const int T = (N + 31) / 32, n_pad = T * 32;
        uint4 *items3; CK(hipMalloc(&items3, ((size_t)n_pad * kRowUnits3 + 64) * 16)); CK(hipMemset(items3, 0, ((size_t)n_pad * kRowUnits3 + 64) * 16));
        k_prep_items3<D><<<(n_pad + 31) / 32, 256>>>(N, n_pad, dQ, dsi, c, items3);
        CK(hipDeviceSynchronize());
        auto kern = k_list3<D, LB_WPS>;
        char name[128]; snprintf(name, sizeof name, "new + LDS-DMA, %d visits ahead, %d waves/SIMD abl=%d", LB_DIST, LB_WPS, LB_ABL);
        run(name, [&]() {
            kern<<<geo.grid1, 512>>>(U, N, ws.users_bf, items3, ws.unorm, qmax_bits, dsu, c, nullptr, 0, geo.ublocks, dtau,
                                      ws.lists, ws.counts, ws.cap, ws.overflow);
        });
        uint32_t *zero_mask; CK(hipMalloc(&zero_mask, (size_t)T * U * 4)); CK(hipMemset(zero_mask, 0, (size_t)T * U * 4));
        auto kern4 = k_list4<D, LB_WPS>;
        snprintf(name, sizeof name, "lean VALU (17-unit rows), %d waves/SIMD abl=%d", LB_WPS, LB_ABL);
        run(name, [&]() {
            kern4<<<geo.grid1, 512>>>(U, N, ws.users_bf, items3, ws.unorm, qmax_bits, dsu, c, zero_mask, 0, geo.ublocks, dtau,
                                       ws.lists, ws.counts, ws.cap, ws.overflow);
        });
#ifdef LB_TRACE
        print_trace();
#endif
#ifndef LB5_UG
#define LB5_UG 2
#define LB5_NW 4
#define LB5_WPS 3
#endif
        {
            auto kern5 = k_list5<D, LB5_UG, LB5_NW, LB5_WPS>;
            const int upb = 32 * LB5_NW * LB5_UG, ub5 = (U + upb - 1) / upb;
            const int resident = 256 * (4 * LB5_WPS / LB5_NW);
            const int grid5 = (int)std::min<long long>(resident, (long long)ub5 * T / 8);
            snprintf(name, sizeof name, "lean, UG=%d NW=%d %d waves/SIMD grid %d abl=%d", LB5_UG, LB5_NW, LB5_WPS, grid5, LB_ABL);
            run(name, [&]() {
                kern5<<<grid5, 64 * LB5_NW>>>(U, N, ws.users_bf, items3, ws.unorm, qmax_bits, dsu, c, zero_mask, 0, ub5, dtau,
                                              ws.lists, ws.counts, ws.cap, ws.overflow);
            });
        }
    }
#ifdef LB_BLKTIME
    {
        static unsigned long long bt[1024][4];
        CK(hipMemcpyFromSymbol(bt, HIP_SYMBOL(g_blk), sizeof(bt)));
        unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0; double sum = 0;
        std::vector<int> per_cu(8 * 64, 0);
        for (int b = 0; b < geo.grid1; ++b) {
            t0 = std::min(t0, bt[b][0]); t1 = std::max(t1, bt[b][1]); sum += (double)(bt[b][1] - bt[b][0]);
            r0 = std::min(r0, bt[b][3]); r1 = std::max(r1, bt[b][3]);
            const unsigned hw = (unsigned)bt[b][2], xcc = (unsigned)(bt[b][2] >> 32) & 15;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_cu[(xcc * 8 + se) * 8 % 512 + 0] += 0; (void)cu; (void)sh;
            per_cu[(xcc & 7) * 64 + ((se & 3) * 16 + sh * 8 + (cu & 7)) % 64] += 1;
        }
        int hist[8] = {0};
        for (int v : per_cu) hist[std::min(v, 7)]++;
        printf("blocks: first start -> last end %.0f cycles, mean block duration %.0f cycles, start spread (100 MHz ticks) %llu; blocks per CU-slot histogram:", (double)(t1 - t0), sum / geo.grid1, r1 - r0);
        for (int k = 0; k < 8; ++k) printf(" %d:%d", k, hist[k]);
        printf("\n  block 0..7 starts/durations:");
        for (int b = 0; b < 8; ++b) printf(" [%llu %llu hw %08x xcc %u]", bt[b][0] - t0, bt[b][1] - bt[b][0], (unsigned)bt[b][2], (unsigned)(bt[b][2] >> 32));
        printf("\n  block 256..259:");
        for (int b = 256; b < 260; ++b) printf(" [%llu %llu hw %08x xcc %u]", bt[b][0] - t0, bt[b][1] - bt[b][0], (unsigned)bt[b][2], (unsigned)(bt[b][2] >> 32));
        printf("\n");
    }
#endif
    return 0;
}
