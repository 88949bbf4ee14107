// filtered_lrelu: bias -> upsampling FIR -> leaky ReLU * gain, clamp -> downsampling FIR in ONE kernel (the intermediate at up x
// resolution lives only in LDS), replacing filtered_lrelu_plugin (torch_utils/ops/filtered_lrelu.{cpp,cu}, filtered_lrelu.py:161-274).
//
//   t[my,mx] = gain1 * sum_k fu'[k] * xz[my + ky - py0, mx + kx - px0]          xz = (x + b) zero-inserted by `up`
//   a        = mode 0: clamp(lrelu(t) * gain)   (and mask := d a / d t, if requested)        mode 1: t * mask
//   y[oy,ox] = gain2 * sum_k fd'[k] * a_pad[oy*down + ky - qy0, ox*down + kx - qx0]          a_pad = a zero-padded / cropped by q
//
// The forward uses q = 0, gain1 = up^2, gain2 = 1 (filtered_lrelu.py:147-150).  The backward is the same kernel in mode 1 with the
// stages transposed (up <-> down, filters swapped and flipped, paddings of upfirdn2d.py:262-269), exactly how the reference chains
// its plugin for gradients (filtered_lrelu.py:240-262) -- but with an fp32 derivative mask instead of packed sign bits.
// Layout: contiguous NCHW (this op is off the EG3D hot path; StyleGAN3-style callers use NCHW).  One block = one 16x16 output tile
// of one (n, c) plane; the input patch and the activated intermediate patch are staged in LDS.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

constexpr int TO = 16;          // output tile edge

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__device__ __forceinline__ int pos_mod_i(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldf<__half>(const __half* p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void stf<__half>(__half* p, float v) { *p = __float2half(v); }

template <typename T>
__global__ void __launch_bounds__(256) filtered_lrelu_kernel(const eg3d_flrelu_params p) {
    extern __shared__ float lds[];
    const int fuh = p.fu ? p.fuh : 1, fuw = p.fu ? p.fuw : 1, fdh = p.fd ? p.fdh : 1, fdw = p.fd ? p.fdw : 1;
    const int Hm = p.H * p.up + p.py0 + p.py1 - (fuh - 1), Wm = p.W * p.up + p.px0 + p.px1 - (fuw - 1);
    const int tiles_x = (p.Wo + TO - 1) / TO;
    const int oy0 = (blockIdx.x / tiles_x) * TO, ox0 = (blockIdx.x % tiles_x) * TO;
    const int c = blockIdx.y, n = blockIdx.z;
    // intermediate patch [TMh][TMw] starting at (my0, mx0); input patch [TXh][TXw] starting at (iy0, ix0)
    const int TMh = (TO - 1) * p.down + fdh, TMw = (TO - 1) * p.down + fdw;
    const int my0 = oy0 * p.down - p.qy0, mx0 = ox0 * p.down - p.qx0;
    const int iy0 = floor_div(my0 - p.py0, p.up), ix0 = floor_div(mx0 - p.px0, p.up);
    const int TXh = (TMh + fuh - 1 + p.up - 1) / p.up + 1, TXw = (TMw + fuw - 1 + p.up - 1) / p.up + 1;
    float* fus = lds;                         // [fuh*fuw]  (gain1 folded, flipped as upfirdn2d does)
    float* fds = fus + fuh * fuw;             // [fdh*fdw]  (gain2 folded)
    float* xs = fds + fdh * fdw;              // [TXh*TXw]
    float* ms = xs + TXh * TXw;               // [TMh*TMw]
    const int tid = threadIdx.x;

    for (int i = tid; i < fuh * fuw; i += 256) {
        const int ky = i / fuw, kx = i % fuw;
        fus[i] = (p.fu ? p.fu[(p.flip_fu ? ky : fuh - 1 - ky) * fuw + (p.flip_fu ? kx : fuw - 1 - kx)] : 1.f) * p.gain1;
    }
    for (int i = tid; i < fdh * fdw; i += 256) {
        const int ky = i / fdw, kx = i % fdw;
        fds[i] = (p.fd ? p.fd[(p.flip_fd ? ky : fdh - 1 - ky) * fdw + (p.flip_fd ? kx : fdw - 1 - kx)] : 1.f) * p.gain2;
    }
    const T* xp = static_cast<const T*>(p.x) + ((int64_t)n * p.C + c) * p.H * p.W;
    const float bias = p.b ? ldf(static_cast<const T*>(p.b) + c) : 0.f;
    for (int i = tid; i < TXh * TXw; i += 256) {
        const int iy = iy0 + i / TXw, ix = ix0 + i % TXw;
        xs[i] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? ldf(xp + (int64_t)iy * p.W + ix) + bias : 0.f;
    }
    __syncthreads();

    // stage 1 + activation: every intermediate sample of the patch
    float* maskp = p.mask ? p.mask + ((int64_t)n * p.C + c) * Hm * Wm : nullptr;
    for (int i = tid; i < TMh * TMw; i += 256) {
        const int my = my0 + i / TMw, mx = mx0 + i % TMw;
        float a = 0.f;
        if ((unsigned)my < (unsigned)Hm && (unsigned)mx < (unsigned)Wm) {
            const int by = my - p.py0, bx = mx - p.px0;
            float t = 0.f;
            for (int ky = pos_mod_i(-by, p.up); ky < fuh; ky += p.up) {
                const int iy = (by + ky) / p.up;                       // exact: (by + ky) % up == 0
                if ((unsigned)iy >= (unsigned)p.H) continue;
                const float* xr = xs + (iy - iy0) * TXw - ix0;
                const float* fr = fus + ky * fuw;
                for (int kx = pos_mod_i(-bx, p.up); kx < fuw; kx += p.up) {
                    const int ix = (bx + kx) / p.up;
                    if ((unsigned)ix < (unsigned)p.W) t += fr[kx] * xr[ix];
                }
            }
            if (p.mode == 0) {
                float d = t < 0.f ? p.slope * p.gain : p.gain;         // lrelu derivative * gain   (bias_act.py:23-33 'lrelu')
                a = t * d;
                if (p.clamp >= 0.f && fabsf(a) > p.clamp) { a = a < 0.f ? -p.clamp : p.clamp; d = 0.f; }
                if (maskp) maskp[(int64_t)my * Wm + mx] = d;           // overlapping patches write identical values
            } else {
                a = t * maskp[(int64_t)my * Wm + mx];
            }
        }
        ms[i] = a;
    }
    __syncthreads();

    // stage 2: one output sample per thread
    const int oy = oy0 + tid / TO, ox = ox0 + tid % TO;
    if (oy < p.Ho && ox < p.Wo) {
        const float* mr = ms + (tid / TO) * p.down * TMw + (tid % TO) * p.down;
        float acc = 0.f;
        for (int ky = 0; ky < fdh; ++ky)
            for (int kx = 0; kx < fdw; ++kx) acc += fds[ky * fdw + kx] * mr[ky * TMw + kx];
        stf(static_cast<T*>(p.y) + (((int64_t)n * p.C + c) * p.Ho + oy) * p.Wo + ox, acc);
    }
}

}  // namespace

extern "C" int eg3d_filtered_lrelu(const eg3d_flrelu_params* pp, void* stream) {
    if (!pp) return EG3D_ERR_INVALID;
    const eg3d_flrelu_params& p = *pp;
    if (!p.x || !p.y || p.N <= 0 || p.C <= 0 || p.H <= 0 || p.W <= 0 || p.up < 1 || p.down < 1) return EG3D_ERR_INVALID;
    if (p.dtype != EG3D_F32 && p.dtype != EG3D_F16) return EG3D_ERR_UNSUPPORTED;
    if ((p.fu && (p.fuh < 1 || p.fuw < 1)) || (p.fd && (p.fdh < 1 || p.fdw < 1))) return EG3D_ERR_INVALID;
    if (p.mode != 0 && p.mode != 1) return EG3D_ERR_INVALID;
    if (p.mode == 1 && !p.mask) return EG3D_ERR_INVALID;
    if (p.mode == 0 && (p.gain <= 0.f || p.slope < 0.f)) return EG3D_ERR_INVALID;          // filtered_lrelu.py:136-137
    const int fuh = p.fu ? p.fuh : 1, fuw = p.fu ? p.fuw : 1, fdh = p.fd ? p.fdh : 1, fdw = p.fd ? p.fdw : 1;
    const int Hm = p.H * p.up + p.py0 + p.py1 - (fuh - 1), Wm = p.W * p.up + p.px0 + p.px1 - (fuw - 1);
    if (Hm < 1 || Wm < 1) return EG3D_ERR_INVALID;
    if (p.Ho != (Hm + p.qy0 + p.qy1 - fdh + p.down) / p.down || p.Wo != (Wm + p.qx0 + p.qx1 - fdw + p.down) / p.down) return EG3D_ERR_INVALID;
    if (p.Ho < 1 || p.Wo < 1) return EG3D_ERR_INVALID;
    if (p.C > 65535 || p.N > 65535) return EG3D_ERR_TOO_LARGE;
    const int TMh = (TO - 1) * p.down + fdh, TMw = (TO - 1) * p.down + fdw;
    const int TXh = (TMh + fuh - 1 + p.up - 1) / p.up + 1, TXw = (TMw + fuw - 1 + p.up - 1) / p.up + 1;
    const size_t smem = sizeof(float) * ((size_t)fuh * fuw + (size_t)fdh * fdw + (size_t)TXh * TXw + (size_t)TMh * TMw);
    if (smem > 64 * 1024) return EG3D_ERR_UNSUPPORTED;                                     // filters / factors far beyond StyleGAN3's
    const dim3 grid(eg3d_cdiv(p.Ho, TO) * eg3d_cdiv(p.Wo, TO), p.C, p.N);
    if (p.dtype == EG3D_F32) hipLaunchKernelGGL(filtered_lrelu_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(filtered_lrelu_kernel<__half>, grid, dim3(256), smem, (hipStream_t)stream, p);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}


// ---- stand-alone activation with the packed sign image (torch_utils/ops/filtered_lrelu.cpp:217-272, filtered_lrelu.cu:1110-1215) ----------
// x (in place, contiguous NCHW fp32 / fp16): v = x * gain;  mode 1 (write signs): v < 0 -> v *= slope, sign 1;  |v| > clamp -> v = +-clamp,
// sign 2;  mode 2 (read signs): the element at (x + sx, y + sy) of the sign image decides: bit 0 -> v *= slope, bit 1 -> v = 0 (elements
// outside the image are only scaled by gain);  mode 0: as mode 1 without writing.  Sign image: [N*C][sH][sW / 4] bytes, 2 bits per element,
// element x in bits 2 (x & 3) of byte x >> 2 (sW a multiple of 4).
namespace {
template <typename T>
__global__ void __launch_bounds__(256) flrelu_act_kernel(T* __restrict__ x, uint8_t* __restrict__ s, int NC, int H, int W, int sH, int sW, int sx, int sy,
                                                         float gain, float slope, float clamp, int mode) {
    // one thread per group of four consecutive elements of a row = one byte of the sign image
    const int wq = (mode == 1 ? sW : ((W + 3) & ~3)) >> 2;
    const int rows = mode == 1 ? sH : H;
    const int64_t total = (int64_t)NC * rows * wq;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q4 = (int)(i % wq);
    const int y = (int)((i / wq) % rows);
    const int64_t q = i / ((int64_t)wq * rows);
    unsigned bits = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int xx = q4 * 4 + e;
        if (xx >= W || y >= H) continue;
        T* pv = x + (q * H + y) * W + xx;
        float v = (float)*pv * gain;
        if (mode == 2) {
            const unsigned ux = (unsigned)(xx + sx), uy = (unsigned)(y + sy);
            if (ux < (unsigned)sW && uy < (unsigned)sH) {
                const unsigned b = (s[(ux >> 2) + (int64_t)(sW >> 2) * (uy + (int64_t)sH * q)] >> ((ux & 3) << 1)) & 3u;
                if (b & 1u) v *= slope;
                if (b & 2u) v = 0.f;
            }
        } else {
            unsigned sg = 0;
            if (v < 0.f) { v *= slope; sg = 1; }
            if (fabsf(v) > clamp) { v = v < 0.f ? -clamp : clamp; sg = 2; }
            bits |= sg << (e << 1);
        }
        *pv = (T)v;
    }
    if (mode == 1) s[q4 + (int64_t)(sW >> 2) * (y + (int64_t)sH * q)] = (uint8_t)bits;
}
}  // namespace

extern "C" int eg3d_filtered_lrelu_act(void* x, uint8_t* signs, int dtype, int NC, int H, int W, int sH, int sW, int sx, int sy, float gain, float slope, float clamp,
                                       int mode, void* stream) {
    if (!x || NC <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return EG3D_ERR_INVALID;
    if (mode != 0 && (!signs || sH <= 0 || sW <= 0 || (sW & 3))) return EG3D_ERR_INVALID;
    if (mode == 1 && (sH < H || sW < W)) return EG3D_ERR_INVALID;
    if (dtype != EG3D_F32 && dtype != EG3D_F16) return EG3D_ERR_UNSUPPORTED;
    if (!(clamp >= 0.f)) clamp = INFINITY;
    const int wq = (mode == 1 ? sW : ((W + 3) & ~3)) >> 2, rows = mode == 1 ? sH : H;
    const int64_t total = (int64_t)NC * rows * wq;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EG3D_F32) hipLaunchKernelGGL(flrelu_act_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)x, signs, NC, H, W, sH, sW, sx, sy, gain, slope, clamp, mode);
    else hipLaunchKernelGGL(flrelu_act_kernel<__half>, dim3(blocks), dim3(256), 0, st, (__half*)x, signs, NC, H, W, sH, sW, sx, sy, gain, slope, clamp, mode);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
