// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two load widths the kernels use (MI355X_MICROARCH.md: the counter
// reports HALF the bytes of a 16 B/lane streaming read; other widths are uncalibrated).  Streams 1 GiB once per launch:
//   k_stream<1>: global_load_dword  (64 lanes x 4 B = one 256-byte row per instruction, the SpMM's gather shape)
//   k_stream<4>: global_load_dwordx4
// run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the counter (KiB) with 1 048 576 KiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int VEC>
__global__ __launch_bounds__(256) void k_stream(const float *__restrict__ src, size_t n_float, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256 * VEC;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < n_float; i += stride) {
        if (VEC == 4) { const v4f v = *reinterpret_cast<const v4f *>(src + i); acc += v.x + v.y + v.z + v.w; }
        else acc += src[i];
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t n = (size_t)1 << 28;            // 1 GiB of floats
    float *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 64);
    hipMemset(src, 0, n * 4);
    for (int r = 0; r < 3; ++r) {
        k_stream<1><<<4096, 256>>>(src, n, out);
        k_stream<4><<<4096, 256>>>(src, n, out);
    }
    hipDeviceSynchronize();
    printf("streamed %zu KiB per launch\n", n * 4 / 1024);
    return 0;
}
