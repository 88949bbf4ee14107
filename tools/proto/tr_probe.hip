// What ds_read_b64_tr_b16 returns: LDS holds halfs h[i] = i; lane l reads at byte address addr(l).  Prints lane -> 4 values.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out, int mode) {
    __shared__ __attribute__((aligned(16))) _Float16 h[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) h[i] = (_Float16)(float)i;
    __syncthreads();
    const int l = threadIdx.x, a = l & 15, g = l >> 4;
    unsigned byte;
    if (mode == 0) byte = l * 8;                                  // lane-linear 8-byte chunks
    else byte = (g * 256 + (a >> 2) * 64 + (a & 3) * 8);          // group g: 4 rows of 32 halfs (64 B pitch), lane a -> row a/4, quad a%4
    auto p = (__attribute__((address_space(3))) hv4*)(uintptr_t)((unsigned)(uintptr_t)h + byte);
    hv4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    f16x4 r = __builtin_bit_cast(f16x4, v);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)r[j];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    float hst[256];
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode); hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, hst[l * 4], hst[l * 4 + 1], hst[l * 4 + 2], hst[l * 4 + 3]);
    }
    return 0;
}
