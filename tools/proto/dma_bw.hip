// Per-workgroup ingest of a weight-streaming kernel on MI355X: how fast can G workgroups each stream S bytes through an LDS ring by LDS-DMA
// (buffer_load ... lds, 16 B / lane) with D KB in flight, and the same through registers (global_load_dwordx4)?  Decides the tile / split
// geometry of the low-resolution convolution (csrc/conv_lr.hip).   hipcc --offload-arch=gfx950 -O3 dma_bw.hip -o dma_bw && ./dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 4 waves; a "step" = 8 KB = 2 DMA instructions per wave; NB-slot ring, NB - 1 steps in flight
template <int NB>
__global__ void __launch_bounds__(256) dma_stream(const char* src, float* out, int steps, int64_t wg_stride, int share) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (int64_t)(share ? blockIdx.x / share : blockIdx.x) * wg_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(steps * 8192), 0x00020000);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    auto issue = [&](int s) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned off = s < steps ? (unsigned)(s * 8192 + (wave * 2 + e) * 1024 + lane * 16) : 0x7ffffff0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)(lds0 + (s & (NB - 1)) * 8192 + (wave * 2 + e) * 1024), 16, off, 0, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < NB - 1; ++s) issue(s);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
        wait_vm<2 * (NB - 2)>();
        __builtin_amdgcn_s_barrier();
        issue(s + NB - 1);
        acc += *reinterpret_cast<const f32x4*>(smem + (s & (NB - 1)) * 8192 + tid * 16);
        acc += *reinterpret_cast<const f32x4*>(smem + (s & (NB - 1)) * 8192 + 4096 + tid * 16);
    }
    wait_vm<0>();
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the same bytes through registers: UNR x 16 B per lane in flight
template <int UNR>
__global__ void __launch_bounds__(256) reg_stream(const char* src, float* out, int steps, int64_t wg_stride, int share) {
    const int tid = threadIdx.x;
    const f32x4* p = reinterpret_cast<const f32x4*>(src + (int64_t)(share ? blockIdx.x / share : blockIdx.x) * wg_stride) + tid;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int n = steps * 2;                 // 4 KB (256 x 16 B) units
    for (int u = 0; u < n; u += UNR) {
        f32x4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) v[k] = __builtin_nontemporal_load(p + (int64_t)(u + k) * 256);
#pragma unroll
        for (int k = 0; k < UNR; ++k) acc += v[k];
    }
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const int64_t TOTAL = 1ll << 30;          // 1 GB source: successive launches walk through it (cold with respect to L2 / the 256 MB MALL)
    char* src; float* out;
    CK(hipMalloc(&src, TOTAL)); CK(hipMalloc(&out, 2048 * 256 * 4));
    CK(hipMemset(src, 1, TOTAL));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)dma_stream<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)dma_stream<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)dma_stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
    printf("%-28s %6s %8s %9s %9s %10s\n", "kernel", "WGs", "KB/WG", "us", "GB/s/WG", "TB/s total");
    for (int share : {0, 4}) {                 // share = 4: groups of four neighbouring workgroups stream the SAME bytes (L2 re-use)
        for (int wgs : {8, 32, 64, 128, 256, 512, 1024}) {
            for (int kb : {36, 288, 1152}) {
                const int steps = kb / 8 + (kb % 8 ? 1 : 0);
                const int64_t stride = (int64_t)steps * 8192;
                const int64_t per_launch = stride * wgs;
                if (per_launch * 6 > TOTAL) continue;
                for (int variant = 0; variant < 5; ++variant) {
                    float best = 1e9f;
                    for (int rep = 0; rep < 5; ++rep) {
                        const char* s = src + ((int64_t)rep * per_launch) % (TOTAL - per_launch);
                        CK(hipEventRecord(e0));
                        switch (variant) {
                            case 0: dma_stream<4><<<wgs, 256, 32768>>>(s, out, steps, stride, share); break;
                            case 1: dma_stream<8><<<wgs, 256, 65536>>>(s, out, steps, stride, share); break;
                            case 2: dma_stream<16><<<wgs, 256, 131072>>>(s, out, steps, stride, share); break;
                            case 3: reg_stream<4><<<wgs, 256>>>(s, out, steps, stride, share); break;
                            default: reg_stream<16><<<wgs, 256>>>(s, out, steps, stride, share); break;
                        }
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0 && ms < best) best = ms;
                    }
                    const char* names[] = {"dma ring 4 (24 KB ahead)", "dma ring 8 (56 KB ahead)", "dma ring 16 (120 KB ahead)", "regs 4 x 16 B / lane", "regs 16 x 16 B / lane"};
                    printf("%-28s %6d %8d %9.1f %9.1f %10.2f %s\n", names[variant], wgs, kb, best * 1e3, stride / (best * 1e-3) / 1e9, per_launch / (best * 1e-3) / 1e12,
                           share ? "(x4 shared)" : "");
                }
            }
        }
    }
    return 0;
}
