// LightGCN normalised-adjacency SpMM for gfx950.
//
// Replaces the tf.sparse_tensor_dense_matmul loop of
// macr_lightgcn/LightGCN.py:297-305 (100 row folds x n_layers) and the
// stack/reduce_mean of :306-307.  A_hat = D^-1/2 A D^-1/2 in CSR
// (macr_lightgcn/utility/load_data.py:112-121), X is the (N,d) embedding table.
//
// One wave per output row; the wave is split into 64/LPR neighbour groups, each
// group streams one neighbour row as LPR lanes x float4 (a coalesced 256-B read at
// d=64), so 4 neighbours are in flight per wave instruction at d=64.  The groups'
// partial rows are combined with cross-lane shuffles; group 0 writes the row.
// HBM-bound: compulsory bytes per layer nnz*8 + (N+1)*4 + 2*N*d*4;
// X itself is L2/Infinity-Cache resident for every real dataset.
#include "common.hpp"

namespace macr {

// Y = A X (if Y), S_out = (S_in + A X) * scale (if S_out).
template <int LPR>
__global__ __launch_bounds__(256) void k_spmm_csr(int N, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const float *__restrict__ val,
                                                  const float *__restrict__ X, float *__restrict__ Y,
                                                  const float *S_in, float *S_out /* may alias (running sum updated in place) */,
                                                  float scale) {
    constexpr int d = 4 * LPR;
    constexpr int NG = kWave / LPR;                 // neighbour groups per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int beg = rowptr[r], end = rowptr[r + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = beg + grp;
    // two neighbours per group in flight to cover L2 latency
    for (; e + NG < end; e += 2 * NG) {
        const int c0 = col[e], c1 = col[e + NG];
        const float a0 = val[e], a1 = val[e + NG];
        const float4 x0 = ld4(X + (size_t)c0 * d + 4 * sub);
        const float4 x1 = ld4(X + (size_t)c1 * d + 4 * sub);
        acc = fma4(a0, x0, acc);
        acc = fma4(a1, x1, acc);
    }
    if (e < end) {
        const float4 x0 = ld4(X + (size_t)col[e] * d + 4 * sub);
        acc = fma4(val[e], x0, acc);
    }
#pragma unroll
    for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
    }
    if (grp == 0) {
        const size_t o = (size_t)r * d + 4 * sub;
        if (Y) st4(Y + o, acc);
        if (S_out) {
            const float4 s = ld4(S_in + o);
            st4(S_out + o, make_float4((s.x + acc.x) * scale, (s.y + acc.y) * scale,
                                       (s.z + acc.z) * scale, (s.w + acc.w) * scale));
        }
    }
}

// out = in * scale   (n_layers == 0 degenerate case) -- float4 per lane
__global__ void k_scale_copy(size_t n_vec, const float *__restrict__ in, float *__restrict__ out, float scale) {
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * blockDim.x) {
        const float4 x = ld4(in + v * 4);
        st4(out + v * 4, make_float4(x.x * scale, x.y * scale, x.z * scale, x.w * scale));
    }
}

int launch_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col, const float *val,
                     const float *E0, float *E, float *work, hipStream_t st) {
    const size_t nd = (size_t)N * d;
    float *bufA = work, *bufB = work + nd;            // alternating layer outputs
    const float inv = 1.0f / (float)(n_layers + 1);
    if (n_layers == 0) {
        k_scale_copy<<<1024, 256, 0, st>>>(nd / 4, E0, E, 1.0f);
        MACR_CHECK_LAUNCH("scale_copy", st);
        return MACR_OK;
    }
    const int grid = (N + 3) / 4;
    const float *X = E0;
    const float *S_in = E0;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        float *Y = last ? nullptr : ((l & 1) ? bufB : bufA);
        // running sum lives in E (S_out); first layer reads E0 as S_in
        MACR_DISPATCH_LPR(d, (k_spmm_csr<LPR><<<grid, 256, 0, st>>>(N, rowptr, col, val, X, Y, S_in, E,
                                                                   last ? inv : 1.0f)));
        MACR_CHECK_LAUNCH("spmm_csr", st);
        X = Y;
        S_in = E;
    }
    return MACR_OK;
}

}  // namespace macr

extern "C" int macr_lgcn_propagate(int N, int d, int n_layers, const int32_t *rowptr, const int32_t *col,
                                   const float *val, const float *E0, float *E, float *work, void *stream) {
    MACR_REQUIRE(N > 0 && n_layers >= 0, MACR_E_INVALID, "lgcn_propagate: N=%d n_layers=%d", N, n_layers);
    MACR_REQUIRE(macr::dim_supported(d), MACR_E_UNSUPPORTED, "lgcn_propagate: d=%d not in {32,64,128,256}", d);
    MACR_REQUIRE(rowptr && col && val && E0 && E && (work || n_layers < 2), MACR_E_INVALID,
                 "lgcn_propagate: null pointer");
    MACR_REQUIRE(E != E0, MACR_E_INVALID, "lgcn_propagate: E must not alias E0");
    return macr::launch_propagate(N, d, n_layers, rowptr, col, val, E0, E, work, macr::as_stream(stream));
}
