// Shared device/host helpers for the gfx950 kernels of the MACR hot path.
// CDNA4 only: wave = 64 lanes, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/macr_hip.h"

namespace macr {

constexpr int kWave = 64;
constexpr int kNumXcd = 8;   // MI355X: block b is dispatched to XCD b % 8 (speed only, never correctness)

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char *fmt, ...);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
// Optional per-kernel timing (macr_timing_*): records a hipEvent on `st` after a launch when enabled.
void timing_mark(const char *name, hipStream_t st);

#define MACR_REQUIRE(cond, code, ...)                 \
    do {                                              \
        if (!(cond)) {                                \
            macr::set_error(__VA_ARGS__);             \
            return (code);                            \
        }                                             \
    } while (0)

#define MACR_CHECK_LAUNCH(what, st)                                                      \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            macr::set_error("%s: %s", (what), hipGetErrorString(e_));                   \
            return MACR_E_LAUNCH;                                                        \
        }                                                                                \
        macr::timing_mark((what), (st));                                                 \
    } while (0)

inline bool dim_supported(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Dispatch on lanes-per-row (d = 4 * LPR floats, one float4 per lane).
#define MACR_DISPATCH_LPR(d, ...)                                  \
    switch (d) {                                                   \
        case 32:  { constexpr int LPR = 8;  __VA_ARGS__; } break;  \
        case 64:  { constexpr int LPR = 16; __VA_ARGS__; } break;  \
        case 128: { constexpr int LPR = 32; __VA_ARGS__; } break;  \
        case 256: { constexpr int LPR = 64; __VA_ARGS__; } break;  \
    }

// ---- device helpers ----------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// (the fused chain written out: left to -ffp-contract, which product stays un-fused depends on the surrounding code, and
// instantiations of one kernel -- k_pair_fwd<LPR, 0/1/2> -- must produce the same bits)
__device__ __forceinline__ float dot4(float4 a, float4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float4 scale4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Sum over the G consecutive lanes of a group (G power of two <= 64); every lane gets the total.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = G >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<kWave>(v); }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// Block-wide sum for <=1024 threads; result valid in thread 0.  scratch: >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float *scratch) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wid] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int k = 0; k < nw; ++k) t += scratch[k];
    return t;
}

// Three block-wide sums behind ONE pair of barriers (the same additions in the same order as three block_sum calls);
// results valid in thread 0.  scratch: >= 48 floats of LDS.
__device__ __forceinline__ void block_sum3(float &a, float &b, float &c, float *scratch) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { scratch[wid] = a; scratch[16 + wid] = b; scratch[32 + wid] = c; }
    __syncthreads();
    float ta = 0.f, tb = 0.f, tc = 0.f;
    if (threadIdx.x == 0)
        for (int k = 0; k < nw; ++k) { ta += scratch[k]; tb += scratch[16 + k]; tc += scratch[32 + k]; }
    a = ta; b = tb; c = tc;
}

// sigmoid as the oracle writes it: 1/(1+exp(-x)), fp32.
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
// d/dx of -log(sigmoid(x)+eps)   and   d/dy of -log((1-sigmoid(y))+eps)   (see oracle/macr_oracle.c)
__device__ __forceinline__ float dneglog_sig(float s, float eps) { return -((s * (1.0f - s)) / (s + eps)); }
__device__ __forceinline__ float dneglog_1msig(float s, float eps) { return (s * (1.0f - s)) / ((1.0f - s) + eps); }

// ---- Adam (shared by the dense pass of train_kernels.hip and the fused epilogue of spmm_kernels.hip) -------------
struct StepScalars {
    float lr_t;          // lr * sqrt(1-beta2^t) / (1-beta1^t)   (TF 1.14 Adam)
    float pad[3];
};

// theta -= (lr_t*m) / (sqrt(v)+eps), one element.  Default: v_sqrt_f32 (<= 1 ulp; denormal v flushes to 0, where eps = 1e-8
// is the whole denominator anyway) and a reciprocal-based quotient with one residual correction (q0 = n*rcp(d),
// q = q0 + (n - d*q0)*rcp(d): correctly rounded in all but rare cases, <= 1 ulp always) -- 13 VALU instructions per
// element instead of the 33 of the IEEE sqrtf and division expansions (scaling for denormals, +-1 ulp candidates,
// v_div_scale/fmas/fixup).  Stand-alone the pass is bound by memory and does not care; riding in the VALU-bound (B,B)
// launch every instruction counts (PMC: the pass was 49 % of that kernel's VALU instructions): 23.1 -> see DESIGN.md.
// The quotient is a step of size ~lr added to theta: 1 ulp of it is ~1e-10, below the resolution of theta itself, and
// tf.train.AdamOptimizer's own rounding is not pinned by anything the reference ships.  -DMACR_ADAM_IEEE restores the
// correctly rounded forms (the oracle uses those; tests compare with tolerances either way).
__device__ __forceinline__ float adam_update(float th, float m, float v, float lr_t, float eps) {
#ifdef MACR_ADAM_IEEE
    return th - (lr_t * m) / (sqrtf(v) + eps);
#else
    const float d = __builtin_amdgcn_sqrtf(v) + eps, n = lr_t * m;
    const float r = __builtin_amdgcn_rcpf(d);
    const float q0 = n * r;
    const float q = fmaf(fmaf(-d, q0, n), r, q0);
    return th - q;
#endif
}

// m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; theta -= lr_t*m/(sqrt(v)+eps)      (TF 1.14 Adam), one element
__device__ __forceinline__ void adam1(float &th, float &m, float &v, float g, float lr_t, float b1, float b2, float eps) {
    m = fmaf(m, b1, g * (1.0f - b1));             // (explicit fused forms: see adam4 in train_kernels.hip)
    v = fmaf(v, b2, (g * g) * (1.0f - b2));
    th = adam_update(th, m, v, lr_t, eps);
}

// ---- test-time score epilogue (macr_mf/model.py:45, :141-142, :199-201) ---------------------------------
// Every operation rounds on its own, in the order the reference's expression evaluates (the empty asm pins the
// product in a register: hipcc would otherwise contract `v - c*s` into one fma -- __fmul_rn does not prevent that
// here, measured as 1-ulp differences against the oracle); oracle/macr_oracle.c::orc_score_epilogue is the same sequence.
__host__ __device__ constexpr bool score_uses_sig_i(int kind) { return kind != MACR_SCORE_NORMAL; }
__host__ __device__ constexpr bool score_uses_sig_u(int kind) {
    return kind == MACR_SCORE_RUBI_BOTH || kind == MACR_SCORE_DIRECT_MINUS_BOTH;
}
template <int KIND>
__device__ __forceinline__ float score_epilogue(float v, float c, float sgi, float su) {
    if (KIND == MACR_SCORE_RUBI_BOTH) { v = v - c; v = v * sgi; v = v * su; }
    else if (KIND == MACR_SCORE_RUBI) { v = v - c; v = v * sgi; }
    else if (KIND == MACR_SCORE_DIRECT_MINUS) { float t = c * sgi; asm volatile("" : "+v"(t)); v = v - t; }
    else if (KIND == MACR_SCORE_DIRECT_MINUS_BOTH) { float t = c * sgi; t = t * su; asm volatile("" : "+v"(t)); v = v - t; }
    return v;
}

// ---- (score,id) ranking keys --------------------------------------------------
// Larger key = ranks earlier: score descending, then id ascending.  0 = empty slot.
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;            // -0.0 ties with +0.0 (they compare equal as floats)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ uint64_t make_key(float score, int32_t id) {
    return (static_cast<uint64_t>(f32_orderable(score)) << 32) | (0xffffffffu - static_cast<uint32_t>(id));
}
__device__ __forceinline__ float key_score(uint64_t k) { return orderable_f32(static_cast<uint32_t>(k >> 32)); }
__device__ __forceinline__ int32_t key_id(uint64_t k) { return static_cast<int32_t>(0xffffffffu - static_cast<uint32_t>(k)); }

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32);
    lo = __shfl_xor(lo, m, kWave);
    hi = __shfl_xor(hi, m, kWave);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// Fill n 32-bit words.  Used instead of hipMemsetAsync everywhere a call sequence may be captured into a HIP graph:
// memset nodes lose their effect from the second replay on (ROCm 7.2, tools/graph_memset_check.py).
static __global__ __launch_bounds__(256) void k_fill_words(uint32_t *__restrict__ p, size_t n, uint32_t value) {
    for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < n; w += (size_t)gridDim.x * blockDim.x) p[w] = value;
}
static inline void fill_words(void *p, size_t n_words, uint32_t value, hipStream_t st) {
    if (!n_words) return;
    size_t g = (n_words + 256 * 16 - 1) / (256 * 16);
    k_fill_words<<<(unsigned)(g < 4096 ? g : 4096), 256, 0, st>>>(static_cast<uint32_t *>(p), n_words, value);
}

// Bitonic sort of one key per lane across the 64 lanes of a wave, DESCENDING by lane index
// (lane 0 ends up with the largest key).
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            const uint64_t other = shfl_xor_u64(key, j);
            const bool upper = (lane & j) != 0;            // I am the higher lane of the pair
            const bool desc = (lane & k) == 0;             // this sub-sequence sorts descending
            // descending: lower lane keeps max.  keep_max = (desc != upper)
            const bool keep_max = (desc != upper);
            const uint64_t mx = key > other ? key : other;
            const uint64_t mn = key > other ? other : key;
            key = keep_max ? mx : mn;
        }
    }
    return key;
}

}  // namespace macr
