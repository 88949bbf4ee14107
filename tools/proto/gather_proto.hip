// Stand-alone prototype: how fast is a gather-ONLY pass over the tri-plane (bilinear, 3 planes, mean) at the sample positions of the
// coarse pass, when nothing else limits the occupancy?  (The fused decode kernels hold 137 VGPRs / 3 waves per SIMD and spend most of their
// time waiting for these loads.)     hipcc -O3 --offload-arch=gfx950 tools/proto/gather_proto.hip -o tools/proto/gather_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
constexpr int FC = 32;
__device__ __forceinline__ void plane_uv(int pl, float x, float y, float z, float& u, float& v) {
    if (pl == 0) { u = x; v = y; } else if (pl == 1) { u = x; v = z; } else { u = z; v = x; }
}
// thread = (sample, channel quad): 12 independent 16-byte loads
template <int TILED>
__global__ void __launch_bounds__(256) gather_kernel(const float* __restrict__ planes, const float4* __restrict__ pos, float4* __restrict__ out, int64_t S,
                                                     int Hp, int Wp, int ldp, float cs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t s = i >> 3;
    const int q = (int)(i & 7);
    if (s >= S) return;
    const float4 p = pos[s];
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        float u, v;
        plane_uv(pl, p.x * cs, p.y * cs, p.z * cs, u, v);
        const float ix = ((u + 1.f) * Wp - 1.f) * 0.5f, iy = ((v + 1.f) * Hp - 1.f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = ix - fx0, wx0 = 1.f - wx1, wy1 = iy - fy0, wy0 = 1.f - wy1;
        const float wts[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int xx = x0 + (c & 1), yy = y0 + (c >> 1);
            const bool ok = (unsigned)xx < (unsigned)Wp && (unsigned)yy < (unsigned)Hp;
            xx = ok ? xx : 0; yy = ok ? yy : 0;
            const float w = ok ? wts[c] : 0.f;
            const float4 t = *reinterpret_cast<const float4*>(planes + ((int64_t)yy * Wp + xx) * ldp + pl * FC + q * 4);
            acc.x += w * t.x; acc.y += w * t.y; acc.z += w * t.z; acc.w += w * t.w;
        }
    }
    const float k = 1.f / 3.f;
    out[s * 8 + q] = make_float4(acc.x * k, acc.y * k, acc.z * k, acc.w * k);
}
int main(int argc, char** argv) {
    const int res = 128, D = 48, Hp = 256, Wp = 256, ldp = 96;
    const int64_t S = (int64_t)res * res * D;
    std::vector<float> hpos(S * 4);
    // pinhole camera at (0, 0, 2.7) looking at the origin, fov 18.8 deg; depths 2.25 .. 3.3, stratified; rays in 32-wide column strips
    const float focal = 4.2647f;
    srand(1);
    int64_t r = 0;
    for (int strip = 0; strip < res / 32; ++strip)
        for (int y = 0; y < res; ++y)
            for (int xs = 0; xs < 32; ++xs, ++r) {
                const int x = strip * 32 + xs;
                const float dx = ((x + 0.5f) / res - 0.5f) / focal, dy = ((y + 0.5f) / res - 0.5f) / focal, dz = -1.f;
                const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
                for (int k = 0; k < D; ++k) {
                    const float t = 2.25f + (3.3f - 2.25f) * (k + (rand() / (float)RAND_MAX)) / D;
                    float* p = &hpos[(r * D + k) * 4];
                    p[0] = dx * inv * t; p[1] = dy * inv * t; p[2] = 2.7f + dz * inv * t; p[3] = t;
                }
            }
    std::vector<float> hpl((size_t)Hp * Wp * ldp);
    for (auto& v : hpl) v = rand() / (float)RAND_MAX - 0.5f;
    float *dpl, *dout; float4* dpos;
    hipMalloc(&dpl, hpl.size() * 4); hipMalloc(&dpos, S * 16); hipMalloc(&dout, S * FC * 4);
    hipMemcpy(dpl, hpl.data(), hpl.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpos, hpos.data(), S * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (int)((S * 8 + 255) / 256);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, dpl, dpos, (float4*)dout, S, Hp, Wp, ldp, 2.f);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, dpl, dpos, (float4*)dout, S, Hp, Wp, ldp, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gather-only: %lld samples, %.1f us per launch (%.2f TB/s of texel reads, %.2f TB/s written)\n", (long long)S, ms * 1e3 / 20,
           S * 12.0 * 128 / (ms / 20 * 1e-3) / 1e12, S * 128.0 / (ms / 20 * 1e-3) / 1e12);
    std::vector<float> ho(64);
    hipMemcpy(ho.data(), dout, 256, hipMemcpyDeviceToHost);
    printf("check %g %g\n", ho[0], ho[33]);
    return 0;
}
