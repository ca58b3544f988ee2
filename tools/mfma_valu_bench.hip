// Micro-benchmark: how many VALU instructions issue for free beside fp32-input MFMAs on gfx950?
// Each wave runs ITER dependent v_mfma_f32_32x32x2_f32 with NV independent v_fma_f32 (4 chains) after each.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_bench tools/mfma_valu_bench.hip && ./mfma_valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, bool MFMA>
__global__ __launch_bounds__(256) void k(float *out, int iters, float x) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = x;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    for (int it = 0; it < iters; ++it) {
        if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV / 4; ++q) {
            v0 = __builtin_fmaf(v0, x, 1.0f); v1 = __builtin_fmaf(v1, x, 1.0f);
            v2 = __builtin_fmaf(v2, x, 1.0f); v3 = __builtin_fmaf(v3, x, 1.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool MFMA>
float run(float *out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, MFMA><<<blocks, 256>>>(out, iters, 0.999f);
    hipEventRecord(e0);
    k<NV, MFMA><<<blocks, 256>>>(out, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4096;
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD (block = 4 waves = one per SIMD)
        const int blocks = 256 * wps;
        printf("waves/SIMD=%d  (cycles per MFMA slot per SIMD at 2.4 GHz)\n", wps);
#define ROW(NV) { float m = run<NV, true>(out, blocks, iters), v = run<NV, false>(out, blocks, iters); \
        printf("  NV=%2d  mfma+valu %.3f ms (%.1f cyc/iter/wave-slot)   valu only %.3f ms (%.1f)\n", NV, m, m * 2.4e6 / iters / wps, v, v * 2.4e6 / iters / wps); }
        ROW(0) ROW(4) ROW(8) ROW(16) ROW(24) ROW(32) ROW(48)
    }
    return 0;
}
