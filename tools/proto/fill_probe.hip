// How long does it take 256 workgroups to each pull the SAME 96 KB (a weight matrix) into LDS?  (torgb_mid_kernel's start-up)
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NB, int MODE>
__global__ void __launch_bounds__(256) fill_kernel(const float4* __restrict__ w, float* out, long long* ticks, int region_f4) {
    extern __shared__ float4 lds[];
    const long long t0 = wall_clock64();
    const float4* src = w;
    if (MODE == 2) src += (size_t)blockIdx.x * region_f4;                 // private region per block
    int rot = 0;
    if (MODE == 1) rot = ((blockIdx.x >> 3) & 31) * (NB * 256 / 32);
    float4 tmp[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int i = j * 256 + threadIdx.x + rot;
        if (i >= NB * 256) i -= NB * 256;
        tmp[j] = src[i];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < NB; ++j) lds[j * 256 + threadIdx.x] = tmp[j];
    __syncthreads();
    const long long t1 = wall_clock64();
    float acc = 0.f;
    for (int j = 0; j < NB; ++j) acc += lds[(j * 256 + threadIdx.x * 7) % (NB * 256)].x;
    if (acc == 12345.f) out[0] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void stream_kernel(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; v.x += 1.f; b[i] = v; }
}
template <int NB, int MODE>
void run(const char* name, const float4* w, float* out, long long* ticks, float4* sa, float4* sb, size_t sn, bool evict, int blocks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<NB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0; double tk = 0;
    std::vector<long long> h(blocks);
    for (int it = 0; it < 12; ++it) {
        if (evict) stream_kernel<<<2048, 256>>>(sa, sb, sn);
        hipEventRecord(e0);
        fill_kernel<NB, MODE><<<blocks, 256, NB * 256 * 16>>>(w, out, ticks, NB * 256);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) { best = ms < best ? ms : best; sum += ms; hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost); double m = 0; for (auto v : h) m += v; tk += m / blocks; }
    }
    printf("%-34s NB=%2d blocks=%d evict=%d: event best %.1f us mean %.1f us; in-kernel fill %.2f us mean per block\n", name, NB, blocks, (int)evict, best * 1e3, sum / 10 * 1e3, tk / 10 * 0.01);
}
int main() {
    const size_t wn = (size_t)512 * 24 * 256;          // float4s: room for a private region per block
    float4 *w, *sa, *sb; float* out; long long* ticks;
    hipMalloc(&w, wn * 16); hipMemset(w, 0, wn * 16);
    const size_t sn = (size_t)8 << 20;                    // 128 MB streams
    hipMalloc(&sa, sn * 16); hipMalloc(&sb, sn * 16); hipMemset(sa, 0, sn * 16);
    hipMalloc(&out, 4); hipMalloc(&ticks, 512 * 8);
    for (int ev = 0; ev < 2; ++ev) {
        run<24, 0>("same 96 KB, same order", w, out, ticks, sa, sb, sn, ev, 256);
        run<24, 1>("same 96 KB, rotated start", w, out, ticks, sa, sb, sn, ev, 256);
        run<24, 2>("private 96 KB per block", w, out, ticks, sa, sb, sn, ev, 256);
        run<12, 0>("same 48 KB, same order", w, out, ticks, sa, sb, sn, ev, 512);
        run<12, 2>("private 48 KB per block", w, out, ticks, sa, sb, sn, ev, 512);
        run<3, 0>("same 12 KB", w, out, ticks, sa, sb, sn, ev, 512);
    }
    return 0;
}
