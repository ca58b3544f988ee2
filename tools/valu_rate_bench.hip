// Issue cost of the VALU instruction classes the (B,B) kernel is made of, on this chip, every SIMD loaded:
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_bench tools/valu_rate_bench.hip && tools/valu_rate_bench
// Per class: a chain-free stream of N instructions per wave (8 independent accumulators), W waves per SIMD, all CUs; the
// s_memtime span of a wave / N = SIMD cycles per wave64 instruction at that occupancy.  Prints one JSON object
// (profiles/r05_valu_rates.json): what `roofline_bxb` in bench.py prices the instruction counts with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float *out, long long *cyc, int iters) {
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.5f + 0.001f * (threadIdx.x + k);
    float b = 1.0001f, c = 0.0003f;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = float2v{a[k], a[k] + 0.25f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(float2v{b, b}), "v"(float2v{c, c}));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
                if (KIND == 3) asm volatile("v_log_f32 %0, %0" : "+v"(a[k]));
                if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(float2v{b, b}));
                if (KIND == 7) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[k]) : "v"(a[(k + 1) & 7]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
double run(int waves_per_simd, int iters, double *wall_us) {
    const int cus = 256, blocks = cus * waves_per_simd;       // 256 threads = 4 waves = one per SIMD
    float *out; long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<KIND><<<blocks, 256>>>(out, cyc, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<KIND><<<blocks, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    *wall_us = 1e3 * ms;
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    hipFree(out); hipFree(cyc);
    return sum / h.size() / (iters * 32.0);                  // counter ticks per instruction of one wave
}

int main() {
    const char *names[8] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_mul_f32", "v_pk_mul_f32", "v_mov_dpp_row_ror"};
    const int iters = 4000;
    printf("{\"note\": \"wall_ns_per_wave_instr_per_simd = kernel time / (instructions per wave * waves per SIMD): the issue cost of one wave64 "
           "instruction in ns with every SIMD of the chip loaded; cycles = that times the 2.4 GHz peak clock\", \"classes\": {");
    for (int w : {1, 2, 4}) {
        double t[8], wall[8];
        t[0] = run<0>(w, iters, &wall[0]); t[1] = run<1>(w, iters, &wall[1]); t[2] = run<2>(w, iters, &wall[2]);
        t[3] = run<3>(w, iters, &wall[3]); t[4] = run<4>(w, iters, &wall[4]); t[5] = run<5>(w, iters, &wall[5]);
        t[6] = run<6>(w, iters, &wall[6]); t[7] = run<7>(w, iters, &wall[7]);
        printf("%s\"waves_per_simd_%d\": {", w == 1 ? "" : ", ", w);
        for (int k = 0; k < 8; ++k) {
            const double ns = 1e3 * wall[k] / (iters * 32.0 * w);
            printf("%s\"%s\": {\"wall_ns_per_wave_instr_per_simd\": %.4f, \"cycles_at_2p4GHz\": %.3f, \"counter_ticks_per_instr\": %.3f}",
                   k ? ", " : "", names[k], ns, ns * 2.4, t[k]);
        }
        printf("}");
    }
    printf("}}\n");
    return 0;
}
