// Deterministic build: the bound-region table, its workspace and the closing pass of a scope (see det.h).  The normal build exports the
// same four entry points; there eg3d_det_enabled() is 0 and eg3d_det_set_workspace() refuses.
#include "det.h"
#include <vector>

namespace {
// every value is added to *target by its own thread (eg3d_det_accumulate: the accumulator's property test)
__global__ void __launch_bounds__(256) det_accumulate_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ target) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) eg3d_acc(target, v[i]);
}
}  // namespace

extern "C" int eg3d_det_accumulate(const float* values, int64_t n, float* target, void* stream) {
    if (!values || !target || n < 0) return EG3D_ERR_INVALID;
    if (n == 0) return EG3D_OK;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, target, 1); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(det_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, values, n, target);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

#if EG3D_DET

namespace {
eg3d_det_host g_host = {nullptr, nullptr, 0, 0};
std::vector<int (*)(eg3d_det_table*)>& tu_setters() { static std::vector<int (*)(eg3d_det_table*)> v; return v; }

struct bind_args { eg3d_det_table t; };
__global__ void det_bind_kernel(eg3d_det_table* dst, const bind_args a) {
    if (threadIdx.x < EG3D_DET_MAXR) dst->r[threadIdx.x] = a.t.r[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) dst->n = a.t.n;
}

struct fin_args { eg3d_det_table t; int first_block[EG3D_DET_MAXR + 1]; };
__global__ void __launch_bounds__(256) det_finalize_kernel(const fin_args a) {
    int r = 0;
    while (r + 1 < a.t.n && (int)blockIdx.x >= a.first_block[r + 1]) ++r;
    const eg3d_det_region R = a.t.r[r];
    const unsigned long long i = (unsigned long long)(blockIdx.x - a.first_block[r]) * 256 + threadIdx.x;
    if (i >= R.count) return;
    long long w[EG3D_DET_NW];
    bool any = false;
#pragma unroll
    for (int k = 0; k < EG3D_DET_NW; ++k) { w[k] = R.words[(unsigned long long)k * R.count + i]; any |= w[k] != 0; }
    if (!any) return;
    // a fixed expression of four exact integers: the same bits whatever order they were accumulated in
    double s = 0.0;
#pragma unroll
    for (int k = EG3D_DET_NW - 1; k >= 0; --k) s += ldexp((double)w[k], k * EG3D_DET_DIGIT - EG3D_DET_OFF);
    R.base[i] += (float)s;
#pragma unroll
    for (int k = 0; k < EG3D_DET_NW; ++k) if (w[k] != 0) R.words[(unsigned long long)k * R.count + i] = 0;
}
}  // namespace

extern "C" void eg3d_det_register_tu(int (*set)(eg3d_det_table*)) { tu_setters().push_back(set); }
extern "C" eg3d_det_host* eg3d_det_host_state() { return &g_host; }

extern "C" void eg3d_det_launch_bind(const eg3d_det_table* t, void* stream) {
    bind_args a; a.t = *t;
    for (int i = t->n; i < EG3D_DET_MAXR; ++i) a.t.r[i] = eg3d_det_region{nullptr, 0, nullptr};
    hipLaunchKernelGGL(det_bind_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_host.table, a);
}

extern "C" void eg3d_det_launch_finalize(const eg3d_det_table* t, void* stream) {
    fin_args a; a.t = *t;
    int nb = 0;
    for (int i = 0; i < t->n; ++i) { a.first_block[i] = nb; nb += (int)((t->r[i].count + 255) / 256); }
    a.first_block[t->n] = nb;
    if (nb) hipLaunchKernelGGL(det_finalize_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
}

extern "C" int eg3d_det_enabled() { return 1; }

extern "C" int64_t eg3d_det_workspace_bytes(int64_t max_elements_per_call) {
    return 4096 + (int64_t)EG3D_DET_NW * 8 * max_elements_per_call;
}

extern "C" int eg3d_det_set_workspace(void* workspace, int64_t bytes, void* stream) {
    if (workspace == nullptr) {             // switch the mode off: every translation unit goes back to float atomics
        for (auto f : tu_setters()) if (f(nullptr)) return 1000;
        g_host.table = nullptr; g_host.words = nullptr; g_host.nwords = 0;
        return EG3D_OK;
    }
    if (bytes < 4096 + 8 * EG3D_DET_NW || ((uintptr_t)workspace & 15)) return EG3D_ERR_INVALID;
    eg3d_zero_words(workspace, bytes / 4, (hipStream_t)stream);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1000;
    g_host.table = reinterpret_cast<eg3d_det_table*>(workspace);
    g_host.words = reinterpret_cast<long long*>(reinterpret_cast<char*>(workspace) + 4096);
    g_host.nwords = (unsigned long long)(bytes - 4096) / 8;
    for (auto f : tu_setters()) if (f(g_host.table)) return 1000;
    return EG3D_OK;
}

extern "C" int eg3d_det_misses(uint32_t* out, void* stream) {
    if (out == nullptr) return EG3D_ERR_INVALID;
    *out = 0;
    if (g_host.table == nullptr) return EG3D_OK;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1000;
    if (hipMemcpy(out, &g_host.table->misses, sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return 1000;
    return EG3D_OK;
}

#else

extern "C" int eg3d_det_enabled() { return 0; }
extern "C" int64_t eg3d_det_workspace_bytes(int64_t) { return 0; }
extern "C" int eg3d_det_set_workspace(void*, int64_t, void*) { return EG3D_ERR_UNSUPPORTED; }
extern "C" int eg3d_det_misses(uint32_t* out, void*) { if (out) *out = 0; return EG3D_OK; }

#endif
