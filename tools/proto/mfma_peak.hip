#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
template <int NACC>
__global__ void __launch_bounds__(256, 2) peak(const f16x8* in, float* out, int iters) {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 7 + i * 13 + blockIdx.x) % 4096];
    for (int i = 0; i < 2; ++i) b[i] = in[(threadIdx.x * 3 + i * 29 + blockIdx.x * 5) % 4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + t) & 3], b[(i ^ t) & 1], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    float scale = argc > 1 ? atof(argv[1]) : 1000.f;
    f16x8* din; float* dout;
    CK(hipMalloc(&din, 4096 * 16)); CK(hipMalloc(&dout, 4096 * 256 * 4));
    _Float16 h[4096 * 8];
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)(scale * ((rand() % 2001) / 1000.f - 1.f));
    CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {512, 1024}) {
        const int iters = 2000;
        peak<8><<<blocks, 256>>>(din, dout, iters);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) peak<8><<<blocks, 256>>>(din, dout, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        double fl = (double)blocks * 4 * iters * 24 * 32768.0;
        printf("scale %g blocks %d: %.1f us  %.0f TFLOP/s\n", scale, blocks, ms * 1e3, fl / ms / 1e9);
    }
    return 0;
}
