// Micro-benchmark (kernel-development helper): what do N short waves cost?  Grid of `blocks` x 256 threads;
// level 0: exit at once; 1: + one dependent scalar load chain (kernarg -> descriptor); 2: + a vector load of a row;
// 3: + a dependent second vector load; 4: + a row store.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int LEVEL>
__global__ __launch_bounds__(256) void k_short(const int4 *__restrict__ items, const float *__restrict__ X, float *Y, int n) {
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (LEVEL == 0) { if (w == 0x7fffffff) Y[0] = 1.f; return; }
    const int4 it = items[w];
    if (LEVEL == 1) { if (it.x == 0x7fffffff) Y[0] = 1.f; return; }
    float a = X[(size_t)it.x * 64 + lane];
    if (LEVEL >= 3) a += X[(size_t)((it.y + (int)a) & (n - 1)) * 64 + lane];
    if (LEVEL >= 4) { Y[(size_t)it.x * 64 + lane] = a; return; }
    if (a == 123.456f) Y[0] = a;
}

template <int LEVEL> void run(int blocks, const int4 *items, const float *X, float *Y, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_short<LEVEL><<<blocks, 256>>>(items, X, Y, n);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) k_short<LEVEL><<<blocks, 256>>>(items, X, Y, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("level %d  blocks %6d (%6d waves)  %7.2f us per launch\n", LEVEL, blocks, blocks * 4, ms * 1e3 / 20);
}

int main() {
    const int n = 65536;
    int4 *items; float *X, *Y;
    hipMalloc(&items, 131072 * sizeof(int4)); hipMalloc(&X, (size_t)n * 64 * 4); hipMalloc(&Y, (size_t)n * 64 * 4);
    int4 *h = (int4 *)malloc(131072 * sizeof(int4));
    for (int i = 0; i < 131072; ++i) { h[i].x = (i * 2654435761u) & (n - 1); h[i].y = (i * 40503u) & (n - 1); h[i].z = h[i].w = 0; }
    hipMemcpy(items, h, 131072 * sizeof(int4), hipMemcpyHostToDevice);
    hipMemset(X, 0, (size_t)n * 64 * 4);
    for (int blocks : {2048, 8192, 19800}) {
        run<0>(blocks, items, X, Y, n); run<1>(blocks, items, X, Y, n); run<2>(blocks, items, X, Y, n);
        run<3>(blocks, items, X, Y, n); run<4>(blocks, items, X, Y, n);
    }
    return 0;
}
