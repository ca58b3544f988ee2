// How global_load_lds_dwordx4 places its data (gfx950): fill 1 KB per wave from a known array, read LDS back.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const int *src, int *out) {
    __shared__ int4 ring[4][3][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    for (int j = 0; j < 3; ++j) {
        const char *p = reinterpret_cast<const char *>(src) + ((size_t)(wid * 3 + j) * 1024) + lane * 16;
        __builtin_amdgcn_global_load_lds((glb_void *)p, (lds_void *)&ring[wid][j][0], 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int j = 0; j < 3; ++j) {
        const int4 q = ring[wid][j][lane];
        int *o = out + ((wid * 3 + j) * 64 + lane) * 4;
        o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
    }
}
int main() {
    const int n = 4 * 3 * 256;
    std::vector<int> h(n), r(n, -1);
    for (int i = 0; i < n; ++i) h[i] = i;
    int *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(o, 0xff, n * 4);
    k<<<1, 256>>>(d, o);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (r[i] != i) { if (bad++ < 16) printf("at %d got %d\n", i, r[i]); }
    printf("%d mismatches of %d\n", bad, n);
    return 0;
}
