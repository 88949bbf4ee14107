// upfirdn2d for gfx950: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate, any strides / dtype.
// Contract: torch_utils/ops/upfirdn2d.py:120-164,169-213 and SURVEY.md Appendix D (behaviour only).
// One thread per output element; the fastest-varying thread index follows the unit-stride axis of x so that both the
// gather and the store are coalesced (NCHW -> along W, channels_last -> along C).  Only the taps that land on a real
// (non-inserted) sample are visited.
#include "common.h"

namespace {

template <typename T> struct AccT { typedef float type; };
template <> struct AccT<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename AccT<T>::type ldv(const T* p, int64_t i) { return (typename AccT<T>::type)p[i]; }
template <> __device__ __forceinline__ float ldv<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ void stv(T* p, int64_t i, typename AccT<T>::type v) { p[i] = (T)v; }
template <> __device__ __forceinline__ void stv<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

struct UpfirdnArgs {
    int N, C, inH, inW, outH, outW, fH, fW;
    int upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
    int64_t xs0, xs1, xs2, xs3, ys0, ys1, ys2, ys3;
};

__device__ __forceinline__ int pos_mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

template <typename T, bool CLAST>
__global__ void __launch_bounds__(256) upfirdn2d_generic(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y, UpfirdnArgs a) {
    typedef typename AccT<T>::type F;
    const int64_t total = (int64_t)a.N * a.C * a.outH * a.outW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int n, c, oy, ox;
        int64_t r = i;
        if (CLAST) { c = (int)(r % a.C); r /= a.C; ox = (int)(r % a.outW); r /= a.outW; oy = (int)(r % a.outH); n = (int)(r / a.outH); }
        else       { ox = (int)(r % a.outW); r /= a.outW; oy = (int)(r % a.outH); r /= a.outH; c = (int)(r % a.C); n = (int)(r / a.C); }
        // upsampled-and-padded coordinate of tap (ky,kx): u = o*down + k - pad0; real sample iff u % up == 0
        const int by = oy * a.downy - a.pady0, bx = ox * a.downx - a.padx0;
        const int ky0 = pos_mod(-by, a.upy), kx0 = pos_mod(-bx, a.upx);
        const T* xp = x + n * a.xs0 + c * a.xs1;
        F acc = 0;
        for (int ky = ky0; ky < a.fH; ky += a.upy) {
            int iy = (by + ky) / a.upy;
            if (by + ky < 0 || iy >= a.inH) continue;
            int fy = a.flip ? ky : a.fH - 1 - ky;
            for (int kx = kx0; kx < a.fW; kx += a.upx) {
                int ix = (bx + kx) / a.upx;
                if (bx + kx < 0 || ix >= a.inW) continue;
                int fx = a.flip ? kx : a.fW - 1 - kx;
                acc += (F)f[fy * a.fW + fx] * ldv<T>(xp, iy * a.xs2 + ix * a.xs3);
            }
        }
        stv<T>(y, n * a.ys0 + c * a.ys1 + oy * a.ys2 + ox * a.ys3, acc * (F)a.gain);
    }
}

// channels-last fp32, float4 over channels, optional accumulate (fused-path resampler)
__global__ void __launch_bounds__(256) upfirdn2d_nhwc4(const float* __restrict__ x, const float* __restrict__ f, float* y, const float* addend,
                                                       int N, int C4, int inH, int inW, int outH, int outW, int fH, int fW, int up,
                                                       int down, int padx0, int pady0, int flip, float gain) {
    __shared__ float fs[64];
    if (threadIdx.x < fH * fW) {
        int ky = threadIdx.x / fW, kx = threadIdx.x % fW;
        fs[threadIdx.x] = f[(flip ? ky : fH - 1 - ky) * fW + (flip ? kx : fW - 1 - kx)] * gain;
    }
    __syncthreads();
    const int64_t total = (int64_t)N * outH * outW * C4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C4);
        int64_t r = i / C4;
        int ox = (int)(r % outW); r /= outW;
        int oy = (int)(r % outH);
        int n = (int)(r / outH);
        const int by = oy * down - pady0, bx = ox * down - padx0;
        const int ky0 = pos_mod(-by, up), kx0 = pos_mod(-bx, up);
        float4 acc = make_float4(0, 0, 0, 0);
        if (up == 1 && fH == 4 && fW == 4) {
            // plain 4x4 FIR (the adjoint filter pass of every up-sampling layer's backward): 16 unconditional loads in flight
            float4 t[16];
            float wg[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int iy = by + (k >> 2), ix = bx + (k & 3);
                const bool ok = (unsigned)iy < (unsigned)inH && (unsigned)ix < (unsigned)inW;
                wg[k] = ok ? fs[k] : 0.f;
                t[k] = x4[((int64_t)(n * inH + (ok ? iy : 0)) * inW + (ok ? ix : 0)) * C4 + c];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc.x += wg[k] * t[k].x; acc.y += wg[k] * t[k].y; acc.z += wg[k] * t[k].z; acc.w += wg[k] * t[k].w; }
        } else
        for (int ky = ky0; ky < fH; ky += up) {
            int iy = (by + ky) / up;
            if (by + ky < 0 || iy >= inH) continue;
            for (int kx = kx0; kx < fW; kx += up) {
                int ix = (bx + kx) / up;
                if (bx + kx < 0 || ix >= inW) continue;
                float w = fs[ky * fW + kx];
                float4 v = x4[((int64_t)(n * inH + iy) * inW + ix) * C4 + c];
                acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            }
        }
        if (addend != nullptr) { const float4 o = reinterpret_cast<const float4*>(addend)[i]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }      // (y itself when accumulating)
        y4[i] = acc;
    }
}

// plain 4x4 FIR (up = down = 1), 2x2 outputs per thread sharing a 5x5 input patch: 25 sixteen-byte loads per 4 outputs instead of 64
__global__ void __launch_bounds__(256) upfirdn2d_nhwc4_fir44(const float* __restrict__ x, const float* __restrict__ f, float* __restrict__ y, int N, int C4,
                                                             int inH, int inW, int outH, int outW, int padx0, int pady0, int flip, float gain,
                                                             int accumulate) {
    __shared__ float fs[16];
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x / 4, kx = threadIdx.x % 4;
        fs[threadIdx.x] = f[(flip ? ky : 3 - ky) * 4 + (flip ? kx : 3 - kx)] * gain;
    }
    __syncthreads();
    const int H2 = (outH + 1) / 2, W2 = (outW + 1) / 2;
    const int64_t total = (int64_t)N * H2 * W2 * C4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        int64_t r = i / C4;
        const int ox0 = (int)(r % W2) * 2; r /= W2;
        const int oy0 = (int)(r % H2) * 2;
        const int n = (int)(r / H2);
        float4 t[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            const int iy = oy0 - pady0 + k / 5, ix = ox0 - padx0 + k % 5;
            const bool ok = (unsigned)iy < (unsigned)inH && (unsigned)ix < (unsigned)inW;
            t[k] = x4[((int64_t)(n * inH + (ok ? iy : 0)) * inW + (ok ? ix : 0)) * C4 + c];
            if (!ok) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int oy = oy0 + dy, ox = ox0 + dx;
                if (oy >= outH || ox >= outW) continue;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const float w = fs[ky * 4 + kx];
                        const float4 v = t[(dy + ky) * 5 + dx + kx];
                        acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
                    }
                const int64_t o = (((int64_t)n * outH + oy) * outW + ox) * C4 + c;
                if (accumulate) { const float4 p = y4[o]; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
                y4[o] = acc;
            }
    }
}

}  // namespace

extern "C" int eg3d_upfirdn2d(const void* x, const float* f, void* y, int dtype, int N, int C, int inH, int inW, const int64_t xs[4],
                              int fH, int fW, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                              int flip, float gain, int outH, int outW, const int64_t ys[4], void* stream) {
    if (!x || !f || !y || !xs || !ys) return EG3D_ERR_INVALID;
    if (N <= 0 || C <= 0 || inH <= 0 || inW <= 0 || fH < 1 || fW < 1) return EG3D_ERR_INVALID;   // upfirdn2d.cpp:28-33
    if (upx < 1 || upy < 1 || downx < 1 || downy < 1) return EG3D_ERR_INVALID;
    if (outW != (inW * upx + padx0 + padx1 - fW + downx) / downx || outH != (inH * upy + pady0 + pady1 - fH + downy) / downy)
        return EG3D_ERR_INVALID;                                                                  // upfirdn2d.cpp:39-40
    if (outW < 1 || outH < 1) return EG3D_ERR_INVALID;
    if ((int64_t)N * C * inH * inW > INT32_MAX || (int64_t)N * C * outH * outW > INT32_MAX) return EG3D_ERR_TOO_LARGE;
    UpfirdnArgs a;
    a.N = N; a.C = C; a.inH = inH; a.inW = inW; a.outH = outH; a.outW = outW; a.fH = fH; a.fW = fW;
    a.upx = upx; a.upy = upy; a.downx = downx; a.downy = downy; a.padx0 = padx0; a.pady0 = pady0; a.flip = flip; a.gain = gain;
    a.xs0 = xs[0]; a.xs1 = xs[1]; a.xs2 = xs[2]; a.xs3 = xs[3]; a.ys0 = ys[0]; a.ys1 = ys[1]; a.ys2 = ys[2]; a.ys3 = ys[3];
    const bool clast = (xs[1] == 1 && C > 1);
    const int64_t total = (int64_t)N * C * outH * outW;
    const int threads = 256;
    int blocks = (int)std::min<int64_t>(eg3d_cdiv(total, threads), 256 * 32);
    hipStream_t st_ = (hipStream_t)stream;
#define LAUNCH_U(T)                                                                                                   \
    do {                                                                                                              \
        if (clast) hipLaunchKernelGGL((upfirdn2d_generic<T, true>), dim3(blocks), dim3(threads), 0, st_, (const T*)x, f, (T*)y, a); \
        else hipLaunchKernelGGL((upfirdn2d_generic<T, false>), dim3(blocks), dim3(threads), 0, st_, (const T*)x, f, (T*)y, a);     \
    } while (0)
    if (dtype == EG3D_F32) LAUNCH_U(float);
    else if (dtype == EG3D_F16) LAUNCH_U(__half);
    else if (dtype == EG3D_F64) LAUNCH_U(double);
    else return EG3D_ERR_UNSUPPORTED;
#undef LAUNCH_U
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

static int upfirdn2d_nhwc_impl(const float* x, const float* f, float* y, const float* addend, int N, int C, int inH, int inW, int fH, int fW, int up,
                               int down, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                               int accumulate, void* stream) {
    if (!x || !f || !y || N <= 0 || C <= 0 || up < 1 || down < 1 || fH < 1 || fW < 1) return EG3D_ERR_INVALID;
    if (addend != nullptr && (accumulate || (up == 1 && down == 1 && fH == 4 && fW == 4) || (reinterpret_cast<uintptr_t>(addend) & 15))) return EG3D_ERR_UNSUPPORTED;
    if (C % 4 != 0 || fH * fW > 64) return EG3D_ERR_UNSUPPORTED;
    if (outW != (inW * up + padx0 + padx1 - fW + down) / down || outH != (inH * up + pady0 + pady1 - fH + down) / down)
        return EG3D_ERR_INVALID;
    if (up == 1 && down == 1 && fH == 4 && fW == 4) {
        const int64_t total4 = (int64_t)N * ((outH + 1) / 2) * ((outW + 1) / 2) * (C / 4);
        const int blocks4 = (int)std::min<int64_t>(eg3d_cdiv(total4, 256), 256 * 16);
        hipLaunchKernelGGL(upfirdn2d_nhwc4_fir44, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, x, f, y, N, C / 4, inH, inW, outH, outW, padx0,
                           pady0, flip, gain, accumulate);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
    const int64_t total = (int64_t)N * outH * outW * (C / 4);
    int blocks = (int)std::min<int64_t>(eg3d_cdiv(total, 256), 256 * 16);
    hipLaunchKernelGGL(upfirdn2d_nhwc4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, f, y, accumulate ? y : addend, N, C / 4, inH, inW, outH, outW, fH,
                       fW, up, down, padx0, pady0, flip, gain);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_upfirdn2d_nhwc(const float* x, const float* f, float* y, int N, int C, int inH, int inW, int fH, int fW, int up,
                                   int down, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                                   int accumulate, void* stream) {
    return upfirdn2d_nhwc_impl(x, f, y, nullptr, N, C, inH, inW, fH, fW, up, down, padx0, padx1, pady0, pady1, flip, gain, outH, outW, accumulate, stream);
}

extern "C" int eg3d_upfirdn2d_nhwc_add(const float* x, const float* f, const float* addend, float* y, int N, int C, int inH, int inW, int fH, int fW, int up,
                                       int down, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW, void* stream) {
    if (!addend) return EG3D_ERR_INVALID;
    return upfirdn2d_nhwc_impl(x, f, y, addend, N, C, inH, inW, fH, fW, up, down, padx0, padx1, pady0, pady1, flip, gain, outH, outW, 0, stream);
}
