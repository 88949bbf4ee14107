// Matrix-pipe probe: a register-only v_mfma_f32_32x32x16_f16 loop on caller-supplied (random) fp16 data.  bench.py times it inside the
// benchmark run, so the "what does the matrix pipe sustain on this chip right now" figure printed next to the conv kernel's roofline
// fraction is MEASURED in that run (the chip clocks to its power budget: MI355X_MICROARCH.md "DVFS give-back"), not a constant.
#include "common.h"

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

namespace {
__global__ void __launch_bounds__(256, 2) mfma_probe_kernel(const pf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    pf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 7 + i * 13 + blockIdx.x) & 4095];
    for (int i = 0; i < 2; ++i) b[i] = in[(threadIdx.x * 3 + i * 29 + blockIdx.x * 5) & 4095];
    pf32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + t) & 3], b[(i ^ t) & 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

// in: 4096 x 8 fp16 (64 KiB); out: blocks x 256 floats.  Executes blocks x 4 waves x iters x 24 MFMAs of 32 x 32 x 16 (32768 flop each).
extern "C" int eg3d_probe_mfma_f16(const void* in, float* out, int blocks, int iters, void* stream) {
    if (!in || !out || blocks < 1 || iters < 1) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const pf16x8*>(in), out, iters);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
