// Micro-benchmark (kernel-development helper): cycles per wave-level vector load instruction on gfx950 for the access
// shapes of a row-gather kernel.  Every CU runs WPC waves; each wave issues ITER x 8 independent loads from a table
// that is L1-resident (16 KB), L2-resident (2 MB) or larger, with per-lane row indices, and folds the data into a sum.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ta_bench tools/ta_bench.hip && tools/ta_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// VEC dwords per lane; GROUP lanes share one row (row = GROUP*VEC*4 bytes); ACTIVE lanes enabled (low lanes)
template <int VEC, int GROUP, int ACTIVE>
__global__ __launch_bounds__(256) void k_ta(const float *__restrict__ tab, uint32_t row_mask, int iters, float *out) {
    const int lane = threadIdx.x & 63;
    const int sub = lane % GROUP;
    uint32_t h = (blockIdx.x * 256 + threadIdx.x) / GROUP * 2654435761u + 12345u;   // same within a lane group
    float acc = 0.f;
    constexpr uint32_t row_bytes = GROUP * VEC * 4;
    if (lane < ACTIVE) {
        for (int it = 0; it < iters; ++it) {
            float part[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                h = h * 1664525u + 1013904223u;
                const uint32_t row = (h >> 8) & row_mask;
                const char *p = reinterpret_cast<const char *>(tab) + row * row_bytes + sub * VEC * 4;
                if (VEC == 4) { v4f v = *reinterpret_cast<const v4f *>(p); part[k] = v.x + v.w; }
                else if (VEC == 2) { v2f v = *reinterpret_cast<const v2f *>(p); part[k] = v.x + v.y; }
                else part[k] = *reinterpret_cast<const float *>(p);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += part[k];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int VEC, int GROUP, int ACTIVE>
void run(const char *name, const float *tab, size_t tab_bytes, float *out, int wpc) {
    const uint32_t row_bytes = GROUP * VEC * 4;
    uint32_t rows = 1; while ((size_t)rows * 2 * row_bytes <= tab_bytes) rows *= 2;
    const int iters = 400, blocks = 256 * wpc / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_ta<VEC, GROUP, ACTIVE><<<blocks, 256>>>(tab, rows - 1, 10, out);
    hipEventRecord(e0);
    k_ta<VEC, GROUP, ACTIVE><<<blocks, 256>>>(tab, rows - 1, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)wpc * iters * 8;
    const double ns_per_load = ms * 1e6 / loads_per_cu;
    const double bytes = (double)ACTIVE * VEC * 4;
    printf("%-34s table %8zu KB  wpc %2d  %7.2f ns/wave-load/CU  (%5.1f cyc @2.1GHz)  %6.1f B/ns/CU  chip %5.2f TB/s\n", name,
           tab_bytes / 1024, wpc, ns_per_load, ns_per_load * 2.1, bytes / ns_per_load, bytes / ns_per_load * 256 / 1e3);
}

int main() {
    float *tab, *out;
    const size_t big = 64u << 20;
    hipMalloc(&tab, big); hipMalloc(&out, 64);
    hipMemset(tab, 0, big);
    for (size_t tb : {(size_t)16 << 10, (size_t)2 << 20, (size_t)16 << 20}) {
        for (int wpc : {8, 16, 32}) {
            run<4, 16, 64>("x4, 4 rows of 256 B, 64 lanes", tab, tb, out, wpc);
        }
        run<4, 16, 32>("x4, 2 rows of 256 B, 32 lanes", tab, tb, out, 32);
        run<4, 16, 16>("x4, 1 row of 256 B, 16 lanes", tab, tb, out, 32);
        run<4, 64, 64>("x4, 1 row of 1 KB, 64 lanes", tab, tb, out, 32);
        run<4, 8, 64>("x4, 8 rows of 128 B, 64 lanes", tab, tb, out, 32);
        run<2, 32, 64>("x2, 2 rows of 256 B, 64 lanes", tab, tb, out, 32);
        run<1, 64, 64>("x1, 1 row of 256 B, 64 lanes", tab, tb, out, 32);
        run<1, 16, 64>("x1, 4 rows of 64 B, 64 lanes", tab, tb, out, 32);
    }
    return 0;
}
