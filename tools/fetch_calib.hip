// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the load shapes the kernels use (MI355X_MICROARCH.md: the counter
// reports HALF the bytes of a 16 B/lane streaming read; other widths are uncalibrated).  Every launch reads 1 GiB once:
//   k_stream<1>: global_load_dword   streaming (64 lanes x 4 B = one 256-byte row per instruction)
//   k_stream<4>: global_load_dwordx4 streaming
//   k_gather<1>: global_load_dword   of 256-byte rows in a pseudo-random order (the SpMM's gather shape: wave = row)
//   k_gather<4>: global_load_dwordx4 of 256-byte rows in a pseudo-random order (16 lanes per row, 4 rows per instruction)
// the gathers visit every row of the 1 GiB table exactly once (an odd multiplier modulo a power of two is a permutation),
// so the bytes are known and nothing can hit a cache.  Run by tools/fetch_calib.sh under `rocprofv3 --pmc FETCH_SIZE`.
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
template <int VEC>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ src, unsigned n_rows, float *out) {
    // rows of 64 floats; a "slot" is one wave-instruction's worth: 1 row (VEC 1) or 4 rows (VEC 4)
    float acc = 0.f;
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = gridDim.x * 4;
    const unsigned rows_per = VEC == 4 ? 4 : 1;
    for (unsigned s = wave; s < n_rows / rows_per; s += n_waves) {
        if (VEC == 4) {
            const unsigned r = ((s * 4 + (lane >> 4)) * 2654435761u) & (n_rows - 1);
            const v4f v = *reinterpret_cast<const v4f *>(src + (size_t)r * 64 + (lane & 15) * 4);
            acc += v.x + v.y + v.z + v.w;
        } else {
            const unsigned r = (s * 2654435761u) & (n_rows - 1);
            acc += src[(size_t)r * 64 + lane];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t n = (size_t)1 << 28;            // 1 GiB of floats
    float *src, *out;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, 64);
    (void)hipMemset(src, 0, n * 4);
    for (int r = 0; r < 3; ++r) {
        k_stream<1><<<4096, 256>>>(src, n, out);
        k_stream<4><<<4096, 256>>>(src, n, out);
        k_gather<1><<<4096, 256>>>(src, (unsigned)(n / 64), out);
        k_gather<4><<<4096, 256>>>(src, (unsigned)(n / 64), out);
    }
    (void)hipDeviceSynchronize();
    printf("read %zu KiB per launch\n", n * 4 / 1024);
    return 0;
}
