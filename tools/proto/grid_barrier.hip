// What does a software grid barrier cost on MI355X, and does a persistent kernel that relies on co-residency replay from a HIP graph?
// (Input for the "one persistent kernel for the 4^2 .. 16^2 blocks" lead)  grid = 256 x {1,2} workgroups of 256 threads, all
// resident; a barrier = release fence + relaxed agent atomic on one counter + spin on a generation word + acquire fence.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier tools/proto/grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(const_cast<unsigned*>(gen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(const_cast<unsigned*>(gen), g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(const_cast<unsigned*>(gen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// two-level form: the blocks of an XCD (block b runs on XCD b % 8) meet on their own counter, the eight last arrivers on a global one
__device__ __forceinline__ void grid_barrier2(unsigned* xcd_count, unsigned* count, volatile unsigned* gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(const_cast<unsigned*>(gen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned* xc = xcd_count + (blockIdx.x & 7) * 32;               // 128 bytes apart
        bool last = false;
        if (__hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks / 8 - 1) {
            __hip_atomic_store(xc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7;
        }
        if (last) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(const_cast<unsigned*>(gen), g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(const_cast<unsigned*>(gen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// iters layers of: every block streams `bytes_per_block` of a weight buffer (different region per layer), adds a partial into out, barrier
template <int LEVELS>
__global__ void __launch_bounds__(256) persistent_kernel(const float4* __restrict__ w, float* __restrict__ out, unsigned* count, unsigned* gen, int iters,
                                                         int f4_per_block) {
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float4* p = w + ((size_t)it * gridDim.x + blockIdx.x) * f4_per_block;
        for (int i = threadIdx.x; i < f4_per_block; i += 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
        if (LEVELS == 2) grid_barrier2(count + 64, count, gen, gridDim.x); else grid_barrier(count, gen, gridDim.x);
    }
    if (acc == 12345.f) out[0] = acc;
}

int main() {
    const int iters = 22;                                    // ~ the layers of the 4^2 .. 16^2 blocks, forward
    for (int levels : {1, 2})
    for (int nb : {256, 512}) {
        for (int kb : {0, 36}) {                             // per-block bytes per layer: 0 (barrier only) or 36 KB (9.4 MB / 256 blocks)
            const int f4 = kb * 1024 / 16;
            float4* w; float* out; unsigned* sync;
            hipMalloc(&w, (size_t)iters * nb * (f4 ? f4 : 1) * 16 + 16); hipMalloc(&out, 4); hipMalloc(&sync, 4096);
            hipMemset(w, 0, (size_t)iters * nb * (f4 ? f4 : 1) * 16 + 16); hipMemset(sync, 0, 4096);
            auto kern = levels == 2 ? persistent_kernel<2> : persistent_kernel<1>;
            hipStream_t st; hipStreamCreate(&st);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, st, w, out, sync, sync + 32, iters, f4);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, st, w, out, sync, sync + 32, iters, f4);
            hipEventRecord(e1, st); hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // the same launch inside a captured graph
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, st, w, out, sync, sync + 32, iters, f4);
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, st); hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int rep = 0; rep < 5; ++rep) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipStreamSynchronize(st);
            float msg; hipEventElapsedTime(&msg, e0, e1);
            printf("%d-level barrier, %d blocks, %2d KB per block and layer, %d layers: %.1f us per launch (%.2f us per layer); from a graph: %.1f us per launch; err %s\n", levels, nb, kb, iters,
                   ms / 50 * 1e3, ms / 50 * 1e3 / iters, msg / 50 * 1e3, hipGetErrorString(hipGetLastError()));
            hipFree(w); hipFree(out); hipFree(sync);
        }
    }
    return 0;
}
