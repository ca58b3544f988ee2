// Micro-benchmark: cost of fp32 global atomics on MI355X for the gradient scatter patterns of pair_bwd.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// mode 0: 16 lanes x float4 per row, 4 agent-scope atomics per lane (current pair_bwd pattern)
// mode 1: plain float4 store
// mode 2: 64 lanes x 1 float per row (one atomic instruction covers the row)
// mode 3: like 0 but workgroup-scope atomics (L2-local)
// mode 4: like 2 but workgroup scope
template <int MODE>
__global__ void k(const int* __restrict__ rows, int n_refs, float* g) {
    const int lane = threadIdx.x & 63;
    if (MODE == 2 || MODE == 4) {
        const int ref = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (ref >= n_refs) return;
        float* p = g + (size_t)rows[ref] * 64 + lane;
        if (MODE == 2) unsafeAtomicAdd(p, 1.0f);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        const int ref = blockIdx.x * 16 + (threadIdx.x >> 4);
        if (ref >= n_refs) return;
        float* p = g + (size_t)rows[ref] * 64 + 4 * (threadIdx.x & 15);
        if (MODE == 0) { unsafeAtomicAdd(p, 1.f); unsafeAtomicAdd(p + 1, 1.f); unsafeAtomicAdd(p + 2, 1.f); unsafeAtomicAdd(p + 3, 1.f); }
        else if (MODE == 1) { *reinterpret_cast<float4*>(p) = make_float4(1, 1, 1, 1); }
        else { for (int q = 0; q < 4; ++q) __hip_atomic_fetch_add(p + q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
}

template <int MODE>
float run(const int* d_rows, int n, float* g, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = (MODE == 2 || MODE == 4) ? (n + 3) / 4 : (n + 15) / 16;
    k<MODE><<<grid, 256>>>(d_rows, n, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k<MODE><<<grid, 256>>>(d_rows, n, g);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main() {
    const int n_rows = 70000, n = 12288;
    float* g; CK(hipMalloc(&g, (size_t)n_rows * 64 * 4)); CK(hipMemset(g, 0, (size_t)n_rows * 64 * 4));
    std::mt19937 rng(1);
    std::vector<int> uni(n), zipf(n), hot(n, 5), uniq(n);
    std::vector<double> cdf(40981); double s = 0; for (int i = 0; i < 40981; ++i) { s += 1.0 / (i + 1); cdf[i] = s; }
    for (int i = 0; i < n; ++i) {
        uni[i] = rng() % n_rows; uniq[i] = i * 5;
        if (i < 4096) { double u = (rng() / 4294967296.0) * s; zipf[i] = 29000 + (int)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()); }
        else zipf[i] = rng() % n_rows;
    }
    int* d; CK(hipMalloc(&d, n * 4));
    struct { const char* name; std::vector<int>* v; } pats[] = {{"unique", &uniq}, {"uniform", &uni}, {"zipf+uniform", &zipf}, {"all-one-row", &hot}};
    for (auto& p : pats) {
        CK(hipMemcpy(d, p.v->data(), n * 4, hipMemcpyHostToDevice));
        printf("%-14s  agent16x4 %7.1f us | store %6.1f us | agent64x1 %7.1f us | wg16x4 %7.1f us | wg64x1 %7.1f us\n", p.name,
               run<0>(d, n, g, 20), run<1>(d, n, g, 20), run<2>(d, n, g, 20), run<3>(d, n, g, 20), run<4>(d, n, g, 20));
    }
    return 0;
}
