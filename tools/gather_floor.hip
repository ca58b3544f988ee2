// Kernel-development probe: what does the gather side of a LightGCN layer cost when NOTHING else is in the way?
// 2.73 M (column, weight) pairs in a plan-owned stream, cut into equal slices, one PERSISTENT wave per slice
// (8 192 waves = every wave slot of the chip, or fewer), lane l owns column l of the accumulator (d = 64), a gather is one
// global_load_dword per lane from a wave-uniform row.  The wave keeps DEPTH gathers in flight in a ROLLING pipeline
// (consume the oldest, issue the next: s_waitcnt vmcnt(DEPTH-1)), the index windows (64 pairs = one coalesced 512-byte
// load) are prefetched one window ahead and handed to the lanes by LDS broadcast reads.  Row ends every ROWLEN entries:
// store the accumulator row, start the next.  Compare with the product's dense layer (42 us) and the pipe floor
// (2.73 M x ~5 cycles / 256 CUs = 22 us).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_floor.hip -o tools/gather_floor && tools/gather_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kWin = 64;          // pairs per window

// stream: uint2 {col | flags, weight bits}; slice w = windows [w * win_per_wave, (w + 1) * win_per_wave)
template <int DEPTH, bool ROWENDS>
__global__ __launch_bounds__(256) void k_floor(const uint2 *__restrict__ stream, int win_per_wave, const float *__restrict__ X,
                                               float *__restrict__ Y, int n_waves) {
    __shared__ uint2 s_win[4][2][kWin];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wid);
    if (w >= n_waves) return;
    const uint2 *src = stream + (size_t)w * win_per_wave * kWin;
    const uint32_t lane_bytes = (uint32_t)lane * 4u;
    float acc = 0.f;
    int out_row = w * 64;                                          // where this wave's rows go (bench: private rows)
    uint2 nxt = src[lane];                                         // window 0
    float x[DEPTH];
    for (int j = 0; j < win_per_wave; ++j) {                       // wave-uniform
        uint2 *win = s_win[wid][j & 1];
        win[lane] = nxt;                                           // (the other buffer may still be read by nobody: one wave owns both)
        if (j + 1 < win_per_wave) nxt = src[(size_t)(j + 1) * kWin + lane];
        // rolling pipeline over the 64 pairs of this window
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const uint32_t c = win[k].x;
            x[k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + ((c & 0xffffffu) * 256u + lane_bytes));
        }
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const uint2 p = win[k];
            acc = fmaf(__builtin_bit_cast(float, p.y), x[k % DEPTH], acc);
            if (k + DEPTH < kWin) {
                const uint32_t c = win[k + DEPTH].x;
                x[k % DEPTH] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + ((c & 0xffffffu) * 256u + lane_bytes));
            }
            if (ROWENDS) {
                if (__builtin_amdgcn_readfirstlane((int)p.x) < 0) {   // bit 31: the row ends here (wave-uniform)
                    Y[(size_t)out_row * 64 + lane] = acc;
                    acc = 0.f; ++out_row;
                }
            }
        }
    }
    Y[(size_t)out_row * 64 + lane] = acc;
}

// the same with the window drained at its end replaced by a pipeline that rolls ACROSS windows (the next window's first
// DEPTH gathers are issued while the current window's last DEPTH are consumed)
template <int DEPTH, bool ROWENDS>
__global__ __launch_bounds__(256) void k_floor_x(const uint2 *__restrict__ stream, int win_per_wave, const float *__restrict__ X,
                                                 float *__restrict__ Y, int n_waves) {
    __shared__ uint2 s_win[4][2][kWin];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wid);
    if (w >= n_waves) return;
    const uint2 *src = stream + (size_t)w * win_per_wave * kWin;
    const uint32_t lane_bytes = (uint32_t)lane * 4u;
    float acc = 0.f;
    int out_row = w * 64;
    s_win[wid][0][lane] = src[lane];
    uint2 nxt = win_per_wave > 1 ? src[kWin + lane] : make_uint2(0u, 0u);
    float x[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
        const uint32_t c = s_win[wid][0][k].x;
        x[k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + ((c & 0xffffffu) * 256u + lane_bytes));
    }
    for (int j = 0; j < win_per_wave; ++j) {
        const uint2 *win = s_win[wid][j & 1];
        uint2 *win_n = s_win[wid][(j + 1) & 1];
        win_n[lane] = nxt;                                         // window j + 1 (garbage-free: zero pairs behind the end)
        nxt = j + 2 < win_per_wave ? src[(size_t)(j + 2) * kWin + lane] : make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const uint2 p = win[k];
            acc = fmaf(__builtin_bit_cast(float, p.y), x[k % DEPTH], acc);
            const uint32_t c = k + DEPTH < kWin ? win[k + DEPTH].x : win_n[k + DEPTH - kWin].x;
            x[k % DEPTH] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(X) + ((c & 0xffffffu) * 256u + lane_bytes));
            if (ROWENDS) {
                if (__builtin_amdgcn_readfirstlane((int)p.x) < 0) {
                    Y[(size_t)out_row * 64 + lane] = acc;
                    acc = 0.f; ++out_row;
                }
            }
        }
    }
    Y[(size_t)out_row * 64 + lane] = acc;
}

int main(int argc, char **argv) {
    const int N = 69716;
    const size_t nnz = 2741710;
    const int rowlen = argc > 1 ? atoi(argv[1]) : 39;
    const int colmod = argc > 2 ? atoi(argv[2]) : N;              // gather from the first colmod rows only (cache-resident probes)
    const int only_waves = argc > 3 ? atoi(argv[3]) : 0;
    const int phases = argc > 4 ? atoi(argv[4]) : 1;              // every wave's slice ordered by source range (P ranges of the table)
    std::mt19937_64 rng(5);
    std::vector<double> cdf(N);
    { double s = 0; for (int k = 0; k < N; ++k) { s += 1.0 / (k + 1); cdf[k] = s; } for (double &c : cdf) c /= s; }
    std::uniform_real_distribution<double> un(0.0, 1.0);
    std::vector<float> T((size_t)N * 64);
    for (float &v : T) v = (float)un(rng) - 0.5f;
    float *dX, *dY; uint2 *dS;
    CK(hipMalloc(&dX, T.size() * 4)); CK(hipMemcpy(dX, T.data(), T.size() * 4, hipMemcpyHostToDevice));
    for (int n_waves : {8192, 4096, 16384}) {
        if (only_waves && n_waves != only_waves) continue;
        const int wpw = (int)((nnz + (size_t)n_waves * kWin - 1) / ((size_t)n_waves * kWin));
        const size_t total = (size_t)n_waves * wpw * kWin;
        std::vector<uint2> st(total);
        std::vector<double> ref((size_t)64, 0.0);
        for (size_t e = 0; e < total; ++e) {
            // half of the gathers Zipf-hot, half uniform (the two sides of the bipartite graph)
            const int c0 = (e & 1) ? (int)std::min<size_t>(std::lower_bound(cdf.begin(), cdf.end(), un(rng)) - cdf.begin(), N - 1) : (int)(un(rng) * N) % N;
            const int c = c0 % colmod;
            float wgt = e < nnz ? 0.01f : 0.f;
            uint32_t cw = (uint32_t)c | ((e % rowlen) == (size_t)rowlen - 1 ? 0x80000000u : 0u);
            st[e] = make_uint2(cw, __builtin_bit_cast(uint32_t, wgt));
        }
        if (phases > 1) {
            const size_t per = (size_t)wpw * kWin;
            for (int w = 0; w < n_waves; ++w)
                std::stable_sort(st.begin() + w * per, st.begin() + (w + 1) * per, [&](const uint2 &a, const uint2 &b) {
                    return (uint64_t)(a.x & 0xffffffu) * phases / N < (uint64_t)(b.x & 0xffffffu) * phases / N; });
        }
        CK(hipMalloc(&dS, total * 8)); CK(hipMemcpy(dS, st.data(), total * 8, hipMemcpyHostToDevice));
        CK(hipMalloc(&dY, (size_t)n_waves * 64 * 64 * 4 + (1 << 20)));
        const int grid = (n_waves + 3) / 4;
        auto time = [&](const char *name, auto launch) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int r = 0; r < 3; ++r) launch();
            CK(hipEventRecord(e0)); for (int r = 0; r < 20; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("waves %5d  windows/wave %2d  %-28s %7.2f us\n", n_waves, wpw, name, ms * 1e3 / 20);
        };
        time("drain/window depth16 norows", [&] { k_floor<16, false><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("drain/window depth32 norows", [&] { k_floor<32, false><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("drain/window depth32 rows", [&] { k_floor<32, true><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("rolling depth16 norows", [&] { k_floor_x<16, false><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("rolling depth32 norows", [&] { k_floor_x<32, false><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("rolling depth16 rows", [&] { k_floor_x<16, true><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("rolling depth32 rows", [&] { k_floor_x<32, true><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        time("rolling depth48 rows", [&] { k_floor_x<48, true><<<grid, 256>>>(dS, wpw, dX, dY, n_waves); });
        CK(hipFree(dS)); CK(hipFree(dY));
    }
    return 0;
}
