// What v_mfma_f32_32x32x16_f16 does with fp16 subnormal INPUTS on gfx950, and its rate against the bf16 instruction.
// The fp16 candidate filter of the evaluator (eval_kernels.hip, k_score_stream_c<.., F16>) states its error bound on
// |x - fp16(x)| <= max(2^-11 |x|, 2^-25): true only if the matrix core does not flush subnormal operands to zero.
//   hipcc --offload-arch=gfx950 -O3 -o tools/f16_mfma_check tools/f16_mfma_check.hip && tools/f16_mfma_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// out[row i of A][col j of B] for one 32x32x16 product: A[i][k] = a_val for k == 0 else 0, B[k][j] = b_val for k == 0
__global__ void k_one(uint16_t a_bits, uint16_t b_bits, float *out) {
    const int lane = threadIdx.x, h = lane >> 5;
    union { uint16_t u[8]; f16x8 v; } a, b;
    for (int k = 0; k < 8; ++k) { a.u[k] = 0; b.u[k] = 0; }
    if (h == 0) { a.u[0] = a_bits; b.u[0] = b_bits; }         // k = 0 lives in lanes 0-31, element 0
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

template <bool F16>
__global__ __launch_bounds__(256) void k_rate(int iters, float *out) {
    union { uint32_t u[4]; f16x8 h; bf16x8 b; } a, b;
    for (int k = 0; k < 4; ++k) { a.u[k] = 0x3c003c00u + threadIdx.x; b.u[k] = 0x3c003c00u; }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        if (F16) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.f) out[0] = s;
}

static uint16_t f16_bits(float f) { _Float16 h = (_Float16)f; uint16_t u; __builtin_memcpy(&u, &h, 2); return u; }

int main() {
    float *d;
    hipMalloc(&d, 64);
    struct { const char *what; uint16_t a, b; double want; } cases[] = {
        {"smallest subnormal 2^-24 x 2^10", 0x0001, f16_bits(1024.f), ldexp(1.0, -14)},
        {"subnormal 0x03ff x 1", 0x03ff, f16_bits(1.f), 1023.0 * ldexp(1.0, -24)},
        {"subnormal x subnormal (2^-24 x 2^-24)", 0x0001, 0x0001, ldexp(1.0, -48)},
        {"smallest normal 2^-14 x 1", 0x0400, f16_bits(1.f), ldexp(1.0, -14)},
        {"largest 65504 x 65504", 0x7bff, 0x7bff, 65504.0 * 65504.0},
    };
    int bad = 0;
    for (auto &c : cases) {
        k_one<<<1, 64>>>(c.a, c.b, d);
        float got;
        hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
        const bool ok = (double)got == c.want;
        bad += !ok;
        printf("%-44s got %.9g want %.9g  %s\n", c.what, got, c.want, ok ? "exact" : "DIFFERENT");
    }
    printf("fp16 subnormal operands of v_mfma_f32_32x32x16_f16: %s\n", bad ? "NOT all preserved" : "preserved (products exact)");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f = 0; f < 2; ++f) {
        const int iters = 20000, blocks = 256 * 8;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (f) k_rate<true><<<blocks, 256>>>(iters, d); else k_rate<false><<<blocks, 256>>>(iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double flops = 2.0 * 32 * 32 * 16 * 2.0 * iters * blocks * 4;
        printf("%s 32x32x16: %.1f TFLOP/s dense\n", f ? "f16 " : "bf16", flops / (best * 1e-3) * 1e-12);
    }
    return bad != 0;
}
