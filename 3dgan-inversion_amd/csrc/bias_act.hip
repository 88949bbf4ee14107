// bias_act for gfx950: y = clamp(act(x + b) * gain) and its first/second-order gradients.
// Behavioural contract: torch_utils/ops/bias_act.py:54-88,128-209 and SURVEY.md Appendix D; written from that
// description (HBM-bound elementwise: 16-byte accesses, grid-stride, bias index from the flat offset).
#include "common.h"

namespace {

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type ld(const T* p, int64_t i) { return (typename Acc<T>::type)p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, typename Acc<T>::type v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

template <typename F>
__device__ __forceinline__ F bias_act_elem(F x, F b, F xref, F yref, F dy, int grad, int act, F alpha, F gain, F clamp) {
    F y;
    if (grad == 0) {
        y = eg3d_act_fwd<F>(x + b, act, alpha) * gain;
        if (clamp >= 0) y = y > clamp ? clamp : (y < -clamp ? -clamp : y);
    } else {
        F yy = yref / gain;
        F xx = xref + b;
        if (grad == 1) y = x * gain * eg3d_act_d1<F>(yy, xx, act, alpha);
        else y = x * dy * gain * eg3d_act_d2<F>(yy, xx, act, alpha);
        if (clamp >= 0 && (yref >= clamp || yref <= -clamp)) y = 0;     // strict-inside passes
    }
    return y;
}

// generic scalar kernel (any dtype / alignment)
template <typename T>
__global__ void __launch_bounds__(256) bias_act_scalar(const T* x, const T* b, const T* xref, const T* yref, const T* dy, T* y,
                                                       int64_t numel, int size_b, int step_b, int grad, int act, float alpha,
                                                       float gain, float clamp) {
    typedef typename Acc<T>::type F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        F bv = b ? ld<T>(b, (i / step_b) % size_b) : (F)0;
        F v = bias_act_elem<F>(ld<T>(x, i), bv, xref ? ld<T>(xref, i) : (F)0, yref ? ld<T>(yref, i) : (F)0,
                               dy ? ld<T>(dy, i) : (F)0, grad, act, (F)alpha, (F)gain, (F)clamp);
        st<T>(y, i, v);
    }
}

// fp32 float4 kernel.  BMODE 0: no bias; 1: bias constant over the 4 elements (step_b % 4 == 0);
// 2: channels-last (step_b == 1, size_b % 4 == 0): bias float4.
template <int BMODE>
__global__ void __launch_bounds__(256) bias_act_vec4(const float4* __restrict__ x, const float* __restrict__ b,
                                                     const float4* __restrict__ xref, const float4* __restrict__ yref,
                                                     const float4* __restrict__ dy, float4* __restrict__ y, int64_t nvec,
                                                     int size_b, int step_b, int grad, int act, float alpha, float gain,
                                                     float clamp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float4 xv = x[i];
        float4 bv = make_float4(0, 0, 0, 0);
        if (BMODE == 1) {
            float s = b[((i * 4) / step_b) % size_b];
            bv = make_float4(s, s, s, s);
        } else if (BMODE == 2) {
            bv = *reinterpret_cast<const float4*>(b + ((i * 4) % size_b));
        }
        float4 xr = xref ? xref[i] : make_float4(0, 0, 0, 0);
        float4 yr = yref ? yref[i] : make_float4(0, 0, 0, 0);
        float4 dv = dy ? dy[i] : make_float4(0, 0, 0, 0);
        float4 o;
        o.x = bias_act_elem<float>(xv.x, bv.x, xr.x, yr.x, dv.x, grad, act, alpha, gain, clamp);
        o.y = bias_act_elem<float>(xv.y, bv.y, xr.y, yr.y, dv.y, grad, act, alpha, gain, clamp);
        o.z = bias_act_elem<float>(xv.z, bv.z, xr.z, yr.z, dv.z, grad, act, alpha, gain, clamp);
        o.w = bias_act_elem<float>(xv.w, bv.w, xr.w, yr.w, dv.w, grad, act, alpha, gain, clamp);
        y[i] = o;
    }
}

inline bool aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int eg3d_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                             int dtype, int64_t numel, int size_b, int step_b, int grad, int act, float alpha, float gain,
                             float clamp, void* stream) {
    if (!x || !y || numel < 0 || grad < 0 || grad > 2 || act < EG3D_ACT_LINEAR || act > EG3D_ACT_SWISH) return EG3D_ERR_INVALID;
    if (numel > INT32_MAX) return EG3D_ERR_TOO_LARGE;      // bias_act.cpp:44
    if (b && (size_b <= 0 || step_b <= 0)) return EG3D_ERR_INVALID;
    /* grad >= 1 with neither xref nor yref is legal (linear: the reference saves nothing, bias_act.py:153-156; refs read as 0) */
    if (grad == 2 && !dy) return EG3D_ERR_INVALID;
    if (numel == 0) return EG3D_OK;
    hipStream_t st_ = (hipStream_t)stream;
    const int threads = 256;
    if (!b) { size_b = 1; step_b = 1; }
    if (dtype == EG3D_F32) {
        bool vec = (numel % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(xref) && aligned16(yref) && aligned16(dy);
        int bmode = 0;
        if (b) {
            if (step_b % 4 == 0) bmode = 1;
            else if (step_b == 1 && size_b % 4 == 0 && aligned16(b)) bmode = 2;
            else vec = false;
        }
        if (vec) {
            int64_t nvec = numel / 4;
            int blocks = (int)std::min<int64_t>(eg3d_cdiv(nvec, threads), 256 * 16);
#define LAUNCH_V(M)                                                                                                     \
    hipLaunchKernelGGL(bias_act_vec4<M>, dim3(blocks), dim3(threads), 0, st_, (const float4*)x, (const float*)b,        \
                       (const float4*)xref, (const float4*)yref, (const float4*)dy, (float4*)y, nvec, size_b, step_b,   \
                       grad, act, alpha, gain, clamp)
            if (bmode == 0) LAUNCH_V(0); else if (bmode == 1) LAUNCH_V(1); else LAUNCH_V(2);
#undef LAUNCH_V
            EG3D_LAUNCH_CHECK();
            return EG3D_OK;
        }
    }
    int blocks = (int)std::min<int64_t>(eg3d_cdiv(numel, threads), 256 * 32);
#define LAUNCH_S(T)                                                                                                    \
    hipLaunchKernelGGL(bias_act_scalar<T>, dim3(blocks), dim3(threads), 0, st_, (const T*)x, (const T*)b, (const T*)xref, \
                       (const T*)yref, (const T*)dy, (T*)y, numel, size_b, step_b, grad, act, alpha, gain, clamp)
    if (dtype == EG3D_F32) LAUNCH_S(float);
    else if (dtype == EG3D_F16) LAUNCH_S(__half);
    else if (dtype == EG3D_F64) LAUNCH_S(double);
    else return EG3D_ERR_UNSUPPORTED;
#undef LAUNCH_S
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_abi_version(void) { return 1; }

extern "C" const char* eg3d_status_string(int status) {
    switch (status) {
        case EG3D_OK: return "ok";
        case EG3D_ERR_INVALID: return "invalid argument";
        case EG3D_ERR_UNSUPPORTED: return "unsupported configuration";
        case EG3D_ERR_TOO_LARGE: return "tensor exceeds int32 indexing";
        case EG3D_ERR_WORKSPACE: return "deterministic build: accumulation targets of the call exceed the workspace lent with eg3d_det_set_workspace (or 40 regions)";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown status";
    }
}
