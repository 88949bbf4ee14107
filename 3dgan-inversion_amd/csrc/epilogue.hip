// Layer epilogues and style/demodulation helpers for gfx950 (NHWC fp32, 16-byte accesses over channels).
//
//  * modconv_epilogue_fwd : [4x4 FIR of the up-2 path] * demod + noise + bias -> activation -> gain -> clamp in ONE pass
//    (the reference runs upfirdn2d, add_, bias_act as three kernels: conv2d_resample.py:129,
//     networks_stylegan2.py:89-90,327-329).
//  * modconv_epilogue_bwd : bias_act gradient (output-keyed, bias_act.cu semantics) + every per-layer reduction the
//    inversion loop needs (bias, demod coefficient, noise_const, noise_strength) in ONE pass over (dout, out).
//  * weight_sqsum / demod_fwd / demod_bwd : the demodulation coefficient d[n,o] = rsqrt(sum (w*s)^2 + 1e-8)
//    (networks_stylegan2.py:62-66) factored as sum_k s^2 * (sum_taps w^2), so that the conv can share weights
//    across the batch (activation-scaled formulation, numerically interchangeable: SURVEY.md section 7).
#include "common.h"
#include "det.h"
#include <cstdlib>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __fp16 ue_fp16x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// PWL: the launch's activation is linear / relu / lrelu (host dispatch): `alpha` then carries the slope and no switch is compiled in
template <bool PWL>
__device__ __forceinline__ float act1(float v, int act, float alpha, float gain, float clamp) {
    v = (PWL ? eg3d_pwl_fwd(v, alpha) : eg3d_act_fwd<float>(v, act, alpha)) * gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

__device__ __forceinline__ void commit_amax(float m, float* out) { eg3d_commit_amax_block(m, out); }        // called by every thread, outside divergent flow
__device__ __forceinline__ float amax4(float m, const float4 v) { return fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }

template <bool PWL>
__global__ void __launch_bounds__(256) epilogue_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int N, int H, int W, int C4,
                                                           int Hz, int Wz, const float* __restrict__ fir, int fh, int fw, int pad0,
                                                           float fir_gain, const float* __restrict__ d, const float* __restrict__ noise,
                                                           int64_t noise_nstride, const float* __restrict__ noise_strength,
                                                           const float* __restrict__ bias, int act, float alpha, float gain, float clamp, float* out_amax) {
    __shared__ float fs[64];
    float am = 0.f;
    if (fir != nullptr && threadIdx.x < fh * fw) {
        int ky = threadIdx.x / fw, kx = threadIdx.x % fw;
        fs[threadIdx.x] = fir[(fh - 1 - ky) * fw + (fw - 1 - kx)] * fir_gain;      // true convolution (flip_filter=False)
    }
    __syncthreads();
    const float strength = noise ? *noise_strength : 0.f;
    const int C = C4 * 4;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int n = (int)(r / H);
        float4 v;
        if (fir != nullptr && fh == 4 && fw == 4) {
            // the on-path case (4x4 FIR after the transposed conv): 16 unconditional loads (clamped address, zero weight outside)
            // so that all of them are in flight together
            float4 t[16];
            float wgt[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int iy = y + (k >> 2) - pad0, ix = x + (k & 3) - pad0;
                const bool ok = (unsigned)iy < (unsigned)Hz && (unsigned)ix < (unsigned)Wz;
                wgt[k] = ok ? fs[k] : 0.f;
                t[k] = ld4(z + ((int64_t)(n * Hz + (ok ? iy : 0)) * Wz + (ok ? ix : 0)) * C + c);
            }
            v = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 16; ++k) { v.x += wgt[k] * t[k].x; v.y += wgt[k] * t[k].y; v.z += wgt[k] * t[k].z; v.w += wgt[k] * t[k].w; }
        } else if (fir != nullptr) {
            v = make_float4(0, 0, 0, 0);
            for (int ky = 0; ky < fh; ++ky) {
                int iy = y + ky - pad0;
                if ((unsigned)iy >= (unsigned)Hz) continue;
                for (int kx = 0; kx < fw; ++kx) {
                    int ix = x + kx - pad0;
                    if ((unsigned)ix >= (unsigned)Wz) continue;
                    float wgt = fs[ky * fw + kx];
                    float4 t = ld4(z + ((int64_t)(n * Hz + iy) * Wz + ix) * C + c);
                    v.x += wgt * t.x; v.y += wgt * t.y; v.z += wgt * t.z; v.w += wgt * t.w;
                }
            }
        } else {
            v = ld4(z + i * 4);
        }
        if (d != nullptr) {
            float4 dv = ld4(d + (int64_t)n * C + c);
            v.x *= dv.x; v.y *= dv.y; v.z *= dv.z; v.w *= dv.w;
        }
        if (noise != nullptr) {
            float nz = noise[(int64_t)n * noise_nstride + (int64_t)y * W + x] * strength;
            v.x += nz; v.y += nz; v.z += nz; v.w += nz;
        }
        if (bias != nullptr) {
            float4 b = ld4(bias + c);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        v.x = act1<PWL>(v.x, act, alpha, gain, clamp); v.y = act1<PWL>(v.y, act, alpha, gain, clamp);
        v.z = act1<PWL>(v.z, act, alpha, gain, clamp); v.w = act1<PWL>(v.w, act, alpha, gain, clamp);
        am = amax4(am, v);
        st4(out + i * 4, v);
    }
    commit_amax(am, out_amax);
}

// 4x4 FIR + demod + noise + bias + activation with a 2x2 output block per thread: the four outputs share a 5x5 input patch, i.e. 25
// sixteen-byte loads instead of 64 (the one-output kernel above is bound by the texture/L1 path: 16 loads per 16 bytes of output).
template <bool PWL>
__global__ void __launch_bounds__(256) epilogue_fwd_fir44_kernel(const float* __restrict__ z, float* __restrict__ out, int N, int H, int W, int C4,
                                                                 int Hz, int Wz, const float* __restrict__ fir, int pad0, float fir_gain,
                                                                 const float* __restrict__ d, const float* __restrict__ noise, int64_t noise_nstride,
                                                                 const float* __restrict__ noise_strength, const float* __restrict__ bias, int act,
                                                                 float alpha, float gain, float clamp, float* out_amax) {
    __shared__ float fs[16];
    float am = 0.f;
    if (threadIdx.x < 16) fs[threadIdx.x] = fir[(3 - threadIdx.x / 4) * 4 + (3 - threadIdx.x % 4)] * fir_gain;      // true convolution
    __syncthreads();
    const float strength = noise ? *noise_strength : 0.f;
    const int C = C4 * 4, H2 = H / 2, W2 = W / 2;
    const int64_t total = (int64_t)N * H2 * W2 * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t r = i / C4;
        const int x0 = (int)(r % W2) * 2; r /= W2;
        const int y0 = (int)(r % H2) * 2;
        const int n = (int)(r / H2);
        float4 t[25];
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            const int iy = y0 - pad0 + k / 5, ix = x0 - pad0 + k % 5;
            const bool ok = (unsigned)iy < (unsigned)Hz && (unsigned)ix < (unsigned)Wz;
            t[k] = ld4(z + ((int64_t)(n * Hz + (ok ? iy : 0)) * Wz + (ok ? ix : 0)) * C + c);
            if (!ok) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 dv = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d != nullptr) dv = ld4(d + (int64_t)n * C + c);
        if (bias != nullptr) bv = ld4(bias + c);
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const float wgt = fs[ky * 4 + kx];
                        const float4 q = t[(oy + ky) * 5 + ox + kx];
                        v.x = fmaf(wgt, q.x, v.x); v.y = fmaf(wgt, q.y, v.y); v.z = fmaf(wgt, q.z, v.z); v.w = fmaf(wgt, q.w, v.w);   // as the reference's CUDA kernel (nvcc contracts v += w * x)
                    }
                const int y = y0 + oy, x = x0 + ox;
                v.x *= dv.x; v.y *= dv.y; v.z *= dv.z; v.w *= dv.w;
                if (noise != nullptr) {
                    const float nz = noise[(int64_t)n * noise_nstride + (int64_t)y * W + x] * strength;
                    v.x += nz; v.y += nz; v.z += nz; v.w += nz;
                }
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                v.x = act1<PWL>(v.x, act, alpha, gain, clamp); v.y = act1<PWL>(v.y, act, alpha, gain, clamp);
                v.z = act1<PWL>(v.z, act, alpha, gain, clamp); v.w = act1<PWL>(v.w, act, alpha, gain, clamp);
                am = amax4(am, v);
                st4(out + (((int64_t)n * H + y) * W + x) * C + c, v);
            }
    }
    commit_amax(am, out_amax);
}

// ---- the up layers' epilogue through LDS, optionally writing the consumer's split operand image ---------------------------------------------
// Same arithmetic as epilogue_fwd_fir44_kernel for a SEPARABLE 4-tap FIR (outer(k, k), the [1,3,3,1] filter of every up layer):
//   phase 1 (lanes along channels, 256-byte runs of a pixel): each thread walks the TH + 3 input rows of its (column, channel quad) once and
//           leaves the TH vertical sums in LDS -- every z element is loaded (TH + 3) / TH x (TW + 3) / TW = 1.6 times instead of 25 / 4 = 6.25;
//   phase 2 (lanes along channels): horizontal sums, demodulation, noise, bias, activation, gain, clamp; 16-byte stores; max|out|;
//   phase 3 (SPLIT only; lanes along pixels): the activated tile, kept in LDS, is multiplied by the CONSUMER's styles and written as that
//           layer's two-piece fp16 operand image (conv_v2.hip layout) -- the separate split pass (read 4 B + write 4 B per element) of the
//           consumer disappears.  The image needs its range before the tile exists, so this form requires clamp >= 0 (the super-resolution
//           head, conv_clamp = 256): |out| <= clamp is the bound; scale = the power of two that brings clamp * max|styles| to [2^13, 2^14).
// Measured on MI355X, 513^2 x 128 -> 512^2 x 128: 94 us (2.9 TB/s) for the 25-load form.
constexpr int UE_TH = 8, UE_TW = 16, UE_CH = 64, UE_COLS = UE_TW + 3;
// LDS pitches (floats).  V (vertical sums, read in phase 2 by 16 lanes per pixel x 4 consecutive pixels = one contiguous 1 KB run per wave-instruction when
// the pitch is exactly 64: conflict-free; with the former 68 the ds_read_b128 lane groups straddled two pixels 4 banks apart -- LdsBankConflict 0.50).
// O (activated tile of the SPLIT form, read in phase 3 with lanes along PIXELS): 68, so that consecutive pixels are 4 banks apart.
#ifndef UE_VPITCH
#define UE_VPITCH 64
#endif
constexpr int UE_PITCH = UE_VPITCH, UE_OPITCH = UE_CH + 4;

__device__ __forceinline__ float ue_range_mul(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, 14 - e);
}

#ifndef UE_XCD_REMAP
#define UE_XCD_REMAP 1
#endif
template <bool SPLIT>
__global__ void __launch_bounds__(256) upconv_epilogue_kernel(const float* __restrict__ z, float* __restrict__ out, int N, int H, int W, int C, int Hz, int Wz,
                                                             float k0, float k1, float k2, float k3, int pad0, float fir_gain, const float* __restrict__ d,
                                                             const float* __restrict__ noise, int64_t noise_nstride, const float* __restrict__ noise_strength,
                                                             const float* __restrict__ bias, float slope, float gain, float clamp, float* out_amax,
                                                             const float* __restrict__ sp_scale_in, _Float16* __restrict__ sp_image, float* sp_scale_out) {
    constexpr int UE_VFLOATS = UE_TH * UE_COLS * UE_PITCH, UE_OFLOATS = UE_TH * UE_TW * UE_OPITCH;
    __shared__ __attribute__((aligned(16))) float V[UE_VFLOATS > UE_OFLOATS ? UE_VFLOATS : UE_OFLOATS];                       // vertical sums [row][column][channel]
    float* const O = V;                  // activated tile (SPLIT): takes V's place once phase 2 has read it (41 KB of LDS: three blocks per CU, not two)
    __shared__ float red[4];
    const int tiles_x = (W + UE_TW - 1) / UE_TW, tiles_y = (H + UE_TH - 1) / UE_TH;
    // (the hardware places block b on XCD b % 8: with the linear order a tile's vertical neighbours -- which re-read three of its input rows --
    //  sit on other XCDs and miss their L2s; remapped, each XCD owns a band of consecutive tile rows)
    const int bid = UE_XCD_REMAP ? eg3d_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int n = bid / (tiles_x * tiles_y);
    const int t = bid - n * tiles_x * tiles_y;
    const int y0 = (t / tiles_x) * UE_TH, x0 = (t % tiles_x) * UE_TW;
    const int c0 = blockIdx.y * UE_CH;
    const float kk[4] = {k3, k2, k1, k0};                    // true convolution: tap i of the window meets filter element 3 - i
    float sp_mul = 1.f;
    if (SPLIT) {         // max|consumer styles| (every block derives it from the same few KB): requested here, reduced behind phase 1's loads, published by phase 1's
                         // barrier -- as a reduction + barrier of its own in front of phase 1 it was a dependent memory round trip per block
        float m = 0.f;
        for (int i = threadIdx.x; i < N * C; i += 256) m = fmaxf(m, fabsf(sp_scale_in[i]));
        sp_mul = m;      // (this thread's partial maximum until the barrier)
    }
    // the noise values of this thread's eight output pixels are requested here, with phase 1's loads: inside the phase-2 loop each was a load
    // under a branch followed by its use -- eight memory round trips in sequence per block
    constexpr int UE_ITEMS = UE_TH * UE_TW * (UE_CH / 4) / 256;
    const float strength = noise ? *noise_strength : 0.f;
    float nzv[UE_ITEMS];
#pragma unroll
    for (int k = 0; k < UE_ITEMS; ++k) {
        const int pix = (threadIdx.x >> 4) + k * 16;
        const int y = y0 + pix / UE_TW, x = x0 + pix % UE_TW;
        const bool okp = noise != nullptr && y < H && x < W;
        nzv[k] = (noise != nullptr ? noise : z)[okp ? (int64_t)n * noise_nstride + (int64_t)y * W + x : 0];        // (a valid address either way: no branch around the load)
        if (!okp) nzv[k] = 0.f;
    }
    // ---- phase 1
    for (int it = threadIdx.x; it < UE_COLS * (UE_CH / 4); it += 256) {
        const int col = it / (UE_CH / 4), c4 = it - col * (UE_CH / 4);
        const int x = x0 - pad0 + col;
        float4 r[UE_TH + 3];
#pragma unroll
        for (int q = 0; q < UE_TH + 3; ++q) {
            const int y = y0 - pad0 + q;
            r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)y < (unsigned)Hz && (unsigned)x < (unsigned)Wz) r[q] = ld4(z + ((int64_t)(n * Hz + y) * Wz + x) * C + c0 + c4 * 4);
        }
#pragma unroll
        for (int oy = 0; oy < UE_TH; ++oy) {
            float4 v;
            v.x = (r[oy].x * kk[0] + r[oy + 1].x * kk[1]) + (r[oy + 2].x * kk[2] + r[oy + 3].x * kk[3]);
            v.y = (r[oy].y * kk[0] + r[oy + 1].y * kk[1]) + (r[oy + 2].y * kk[2] + r[oy + 3].y * kk[3]);
            v.z = (r[oy].z * kk[0] + r[oy + 1].z * kk[1]) + (r[oy + 2].z * kk[2] + r[oy + 3].z * kk[3]);
            v.w = (r[oy].w * kk[0] + r[oy + 1].w * kk[1]) + (r[oy + 2].w * kk[2] + r[oy + 3].w * kk[3]);
            *reinterpret_cast<float4*>(V + (oy * UE_COLS + col) * UE_PITCH + c4 * 4) = v;
        }
    }
    if (SPLIT) {
        float m = sp_mul;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (SPLIT) {
        sp_mul = ue_range_mul(clamp * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *sp_scale_out = sp_mul;
    }
    // ---- phase 2: item = (row, column, channel quad); a thread keeps its channel quad
    const int c4 = threadIdx.x & 15, c = c0 + c4 * 4;
    float4 dv = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d != nullptr) dv = ld4(d + (int64_t)n * C + c);
    if (bias != nullptr) bv = ld4(bias + c);
    float am = 0.f;
    float4 keep[SPLIT ? UE_TH * UE_TW * (UE_CH / 4) / 256 : 1];
#pragma unroll
    for (int k = 0; k < UE_TH * UE_TW * (UE_CH / 4) / 256; ++k) {
        const int pix = (threadIdx.x >> 4) + k * 16;                                  // 0 .. 127
        const int oy = pix / UE_TW, ox = pix - oy * UE_TW;
        const int y = y0 + oy, x = x0 + ox;
        const float* vp = V + (oy * UE_COLS + ox) * UE_PITCH + c4 * 4;
        const float4 a0 = *reinterpret_cast<const float4*>(vp), a1 = *reinterpret_cast<const float4*>(vp + UE_PITCH),
                     a2 = *reinterpret_cast<const float4*>(vp + 2 * UE_PITCH), a3 = *reinterpret_cast<const float4*>(vp + 3 * UE_PITCH);
        float4 v;
        v.x = ((a0.x * kk[0] + a1.x * kk[1]) + (a2.x * kk[2] + a3.x * kk[3])) * fir_gain;
        v.y = ((a0.y * kk[0] + a1.y * kk[1]) + (a2.y * kk[2] + a3.y * kk[3])) * fir_gain;
        v.z = ((a0.z * kk[0] + a1.z * kk[1]) + (a2.z * kk[2] + a3.z * kk[3])) * fir_gain;
        v.w = ((a0.w * kk[0] + a1.w * kk[1]) + (a2.w * kk[2] + a3.w * kk[3])) * fir_gain;
        const bool ok = y < H && x < W;
        const float nz = nzv[k] * strength;
        v.x = act1<true>(v.x * dv.x + nz + bv.x, 0, slope, gain, clamp); v.y = act1<true>(v.y * dv.y + nz + bv.y, 0, slope, gain, clamp);
        v.z = act1<true>(v.z * dv.z + nz + bv.z, 0, slope, gain, clamp); v.w = act1<true>(v.w * dv.w + nz + bv.w, 0, slope, gain, clamp);
        if (ok) {
            am = amax4(am, v);
            st4(out + (((int64_t)n * H + y) * W + x) * C + c, v);
        }
        if (SPLIT) keep[k] = v;
    }
    if (SPLIT) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UE_TH * UE_TW * (UE_CH / 4) / 256; ++k)
            *reinterpret_cast<float4*>(O + ((threadIdx.x >> 4) + k * 16) * UE_OPITCH + c4 * 4) = keep[k];
        // ---- phase 3: item = (octet, pixel), lanes along the pixels of a tile row (16 x 16 bytes contiguous in a plane's row)
        __syncthreads();
        const int noct = C / 8;
        const int64_t HW = (int64_t)H * W;
        f16x8_t* img = reinterpret_cast<f16x8_t*>(sp_image);
#pragma unroll
        for (int k = 0; k < UE_TH * UE_TW * (UE_CH / 8) / 256; ++k) {
            const int item = threadIdx.x + k * 256;
            const int pix = item & (UE_TH * UE_TW - 1), kl = item >> 7;
            const int oy = pix / UE_TW, ox = pix - oy * UE_TW;
            const int y = y0 + oy, x = x0 + ox;
            if (y >= H || x >= W) continue;
            const float4 lo = *reinterpret_cast<const float4*>(O + pix * UE_OPITCH + kl * 8), hi = *reinterpret_cast<const float4*>(O + pix * UE_OPITCH + kl * 8 + 4);
            const float* sr = sp_scale_in + (int64_t)n * C + c0 + kl * 8;
            const float v[8] = {lo.x * sr[0], lo.y * sr[1], lo.z * sr[2], lo.w * sr[3], hi.x * sr[4], hi.y * sr[5], hi.z * sr[6], hi.w * sr[7]};
            f16x8_t h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = v[2 * q] * sp_mul, b = v[2 * q + 1] * sp_mul;
                const ue_fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(a, b);
                const float ra = __builtin_amdgcn_fmed3f((a - (float)hh[0]) * 2048.f, -65504.f, 65504.f);
                const float rb = __builtin_amdgcn_fmed3f((b - (float)hh[1]) * 2048.f, -65504.f, 65504.f);
                h[2 * q] = (_Float16)hh[0]; h[2 * q + 1] = (_Float16)hh[1];
                l[2 * q] = (_Float16)ra; l[2 * q + 1] = (_Float16)rb;
            }
            const int ko = c0 / 8 + kl;
            img[((int64_t)(n * 2 + 0) * noct + ko) * HW + (int64_t)y * W + x] = h;
            img[((int64_t)(n * 2 + 1) * noct + ko) * HW + (int64_t)y * W + x] = l;
        }
    }
    commit_amax(am, out_amax);
}

// derivative factor and recovered pre-activation for one element
// (inv_gain = 1 / gain, inv_alpha = 1 / alpha, formed once per thread: two IEEE divisions per element -- ~20 instructions -- were half of
//  the arithmetic of these passes; the products differ from the quotients by at most one ulp, in `pre` only -- signs decide everything else)
template <bool PWL>
__device__ __forceinline__ void bwd1(float dout, float o, int act, float alpha, float gain, float clamp, float inv_gain, float inv_alpha, float& dy, float& pre) {
    float yy = o * inv_gain;
    float g = dout * gain * (PWL ? eg3d_pwl_d1(yy, alpha) : eg3d_act_d1<float>(yy, 0.f, act, alpha));
    if (clamp >= 0.f && (o >= clamp || o <= -clamp)) g = 0.f;
    dy = g;
    if (PWL) pre = (yy > 0.f || alpha == 0.f) ? yy : yy * inv_alpha;        // alpha = slope: 1 (linear), alpha (lrelu); relu is not invertible
    else pre = (act == EG3D_ACT_LRELU) ? (yy > 0.f ? yy : yy * inv_alpha) : yy;     // linear / lrelu are invertible
}

// grid = (blocks_x, N).  Block: EPI_BWD_THREADS threads = PPB pixels x C4 channel-quads.  Big blocks on purpose: every block ends with
// 2*C atomics onto the same dbias / dd addresses (measured: launch time grows linearly with the block count), so the
// parallelism comes from 16 waves per block rather than from many blocks.
constexpr int EPI_BWD_THREADS = 1024;
// blocks of the activation-backward / finishing passes: one block of 16 waves per CU (tuned in round 3; was the EG3D_EPI_BWD_CAP knob)
static int epi_bwd_cap() { return 256; }
// FIN: the incoming gradient is not read but formed on the fly as the finish of a split-K data gradient of the CONSUMER layer,
// dout = fz * fs[n,c] (+ fadd), and that layer's style gradient fds[n,c] += sum_px fz * out rides along (eg3d_dgrad_finish_act).
struct FinArgs { const float* z; const float* s; const float* addend; float* ds; const float* dy4; const float* wa4;
                 // SPLIT: dz also (or only: dz null) leaves as the two-piece fp16 operand image of the data gradient that consumes it
                 uint2* simg; const float* dy_amax; const float* add_amax; float* sp_scale; };
// (dy4 / wa4, with z null: the consumer is a 1x1 layer with four (padded) outputs -- toRGB of the super-resolution head -- whose data gradient
//  z[c] = sum_o dy[px][o] wa[c][o] is four multiply-adds per element: formed here instead of by a GEMM launch with a 4-deep contraction,
//  eg3d_torgb_dgrad_act)
// SPLIT (with FIN + dy4): the split pass that used to follow needs max|dz| before it can write a single piece.  A BOUND does as well -- the
// pieces are floating point, a scale that is a few octaves too cautious costs range at the bottom, not precision: |dz| <= gain * (max|dy| *
// max_{n,c}(sum_o |wa[c][o]|) |s[n,c]| |d[n,c]|) + max|addend| * max |d|) since |act'| <= 1.  Every block derives it from the same few
// hundred numbers.  512^2 x 128: 85 + 52 us (this pass + split pass) -> 61 us, and no fp32 dz at all when nobody else reads it.
// barrier that orders LDS traffic only (__syncthreads also drains the vector-memory counter: it would wait for prefetched global loads)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool PWL, bool FIN, bool SPLIT = false>
__global__ void __launch_bounds__(EPI_BWD_THREADS) epilogue_bwd_kernel(const FinArgs fin, const float* __restrict__ dout, const float* __restrict__ outv, float* __restrict__ dz,
                                                           int H, int W, int C4, const float* __restrict__ d, const float* __restrict__ noise,
                                                           int64_t noise_nstride, const float* __restrict__ noise_strength,
                                                           const float* __restrict__ bias, int act, float alpha, float gain, float clamp,
                                                           float* __restrict__ dbias, float* __restrict__ dd, float* __restrict__ dnoise,
                                                           int64_t dnoise_nstride, float* __restrict__ dstrength, float* __restrict__ dz_amax) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [PPB][C4][8] floats + 1
    const int n = blockIdx.y;
    const int C = C4 * 4;
    const int ppb = EPI_BWD_THREADS / C4 > 0 ? EPI_BWD_THREADS / C4 : 1;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const bool active = pl < ppb;
    const int HW = H * W;
    const float strength = noise ? *noise_strength : 0.f;
    const float inv_gain = 1.0f / gain, inv_alpha = alpha != 0.f ? 1.0f / alpha : 0.f;
    const int c = c4 * 4;
    float4 dv = make_float4(1, 1, 1, 1), bv = make_float4(0, 0, 0, 0);
    if (active && d) dv = ld4(d + (int64_t)n * C + c);
    if (active && bias) bv = ld4(bias + c);
    float4 accb = make_float4(0, 0, 0, 0), accd = make_float4(0, 0, 0, 0), accz = make_float4(0, 0, 0, 0), fsv = make_float4(1, 1, 1, 1);
    if (FIN && active && fin.s) fsv = ld4(fin.s + (int64_t)n * C + c);
    float4 w40 = make_float4(0, 0, 0, 0), w41 = w40, w42 = w40, w43 = w40;
    if (FIN && active && fin.dy4) { w40 = ld4(fin.wa4 + (c + 0) * 4); w41 = ld4(fin.wa4 + (c + 1) * 4); w42 = ld4(fin.wa4 + (c + 2) * 4); w43 = ld4(fin.wa4 + (c + 3) * 4); }
    float accs = 0.f, amax = 0.f;
    float sp_mul = 1.f;
    if constexpr (SPLIT) {
        float m1 = 0.f, m2 = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.y * C; i += EPI_BWD_THREADS) {
            const int cc = i % C;
            const float4 wr = ld4(fin.wa4 + cc * 4);
            const float dval = d ? fabsf(d[i]) : 1.f;
            m1 = fmaxf(m1, (fabsf(wr.x) + fabsf(wr.y) + fabsf(wr.z) + fabsf(wr.w)) * (fin.s ? fabsf(fin.s[i]) : 1.f) * dval);
            m2 = fmaxf(m2, dval);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { m1 = fmaxf(m1, __shfl_xor(m1, o)); m2 = fmaxf(m2, __shfl_xor(m2, o)); }
        if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = m1; red[(threadIdx.x >> 6) * 2 + 1] = m2; }
        __syncthreads();
        m1 = 0.f; m2 = 0.f;
        for (int w = 0; w < EPI_BWD_THREADS / 64; ++w) { m1 = fmaxf(m1, red[w * 2]); m2 = fmaxf(m2, red[w * 2 + 1]); }
        __syncthreads();
        const float bound = gain * (*fin.dy_amax * m1 + (fin.add_amax != nullptr ? *fin.add_amax * m2 : 0.f));
        sp_mul = ue_range_mul(bound);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *fin.sp_scale = sp_mul;
    }
    // lanes of one pixel are contiguous; groups of min(C4,64) lanes can be shuffle-reduced when C4 is a power of two
    const bool pow2 = (C4 & (C4 - 1)) == 0;
    const int grp = C4 < 64 ? C4 : 64;
    if (active) {
        // U pixels per trip: all 2*U 16-byte loads are in flight before the first dependent instruction (the kernel is a pure
        // stream over dout/out/dz; with one pixel per trip each wave had two loads outstanding and 8 waves/CU could not cover
        // the HBM latency).
        constexpr int U = SPLIT ? 2 : 4;          // (SPLIT holds two trips in registers: the one being processed and the prefetched one)
        const int stride = gridDim.x * ppb;
        // (the trip count is block-uniform -- `base` -- because the SPLIT form has barriers inside the loop)
        // SPLIT: the next trip's loads are issued before this trip's arithmetic / staging / stores (the two barriers of the staging would
        // otherwise leave every wave of the block without a load in flight twice per trip; they wait on LDS only -- lds_barrier -- so the
        // prefetch stays outstanding across them)
        struct Raw { float4 o[U], g[U], av[U]; float nraw[U]; };      // g: dout | the split-K sum z | the four toRGB gradients of the pixel
        auto load_raw = [&](const int base, Raw& r) {
            const int pix0 = base + pl;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = pix0 + u * stride;
                const bool ok = pix < HW;
                const int64_t off = ((int64_t)n * HW + (ok ? pix : base)) * C + c;
                r.o[u] = ld4(outv + off);
                if (FIN) {
                    r.g[u] = fin.dy4 ? ld4(fin.dy4 + ((int64_t)n * HW + (ok ? pix : base)) * 4) : ld4(fin.z + off);
                    r.av[u] = fin.addend ? ld4(fin.addend + off) : make_float4(0, 0, 0, 0);
                } else {
                    r.g[u] = ld4(dout + off);
                }
                r.nraw[u] = (noise && ok) ? noise[(int64_t)n * noise_nstride + pix] : 0.f;
            }
        };
        Raw cur;
        const int first = blockIdx.x * ppb;
        if (SPLIT && first < HW) load_raw(first, cur);
        for (int base = first; base < HW; base += U * stride) {
            const int pix0 = base + pl;
            Raw nxt;
            if constexpr (SPLIT) { if (base + U * stride < HW) load_raw(base + U * stride, nxt); }
            else load_raw(base, cur);
            float4 g[U], o[U];
            float nraw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = pix0 + u * stride;
                const bool ok = pix < HW;
                o[u] = cur.o[u];
                if (FIN) {
                    float4 zz = cur.g[u];
                    if (fin.dy4) {
                        const float4 g4 = cur.g[u];
                        zz.x = fmaf(g4.w, w40.w, fmaf(g4.z, w40.z, fmaf(g4.y, w40.y, g4.x * w40.x)));
                        zz.y = fmaf(g4.w, w41.w, fmaf(g4.z, w41.z, fmaf(g4.y, w41.y, g4.x * w41.x)));
                        zz.z = fmaf(g4.w, w42.w, fmaf(g4.z, w42.z, fmaf(g4.y, w42.y, g4.x * w42.x)));
                        zz.w = fmaf(g4.w, w43.w, fmaf(g4.z, w43.z, fmaf(g4.y, w43.y, g4.x * w43.x)));
                    }
                    const float4 av = cur.av[u];
                    if (ok && fin.ds) { accz.x += zz.x * o[u].x; accz.y += zz.y * o[u].y; accz.z += zz.z * o[u].z; accz.w += zz.w * o[u].w; }
                    g[u] = make_float4(zz.x * fsv.x + av.x, zz.y * fsv.y + av.y, zz.z * fsv.z + av.z, zz.w * fsv.w + av.w);
                } else {
                    g[u] = cur.g[u];
                }
                nraw[u] = cur.nraw[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = pix0 + u * stride;
                if (pix >= HW) break;
                const int64_t off = ((int64_t)n * HW + pix) * C + c;
                float4 dy, pre;
                bwd1<PWL>(g[u].x, o[u].x, act, alpha, gain, clamp, inv_gain, inv_alpha, dy.x, pre.x); bwd1<PWL>(g[u].y, o[u].y, act, alpha, gain, clamp, inv_gain, inv_alpha, dy.y, pre.y);
                bwd1<PWL>(g[u].z, o[u].z, act, alpha, gain, clamp, inv_gain, inv_alpha, dy.z, pre.z); bwd1<PWL>(g[u].w, o[u].w, act, alpha, gain, clamp, inv_gain, inv_alpha, dy.w, pre.w);
                const float4 zz = make_float4(dy.x * dv.x, dy.y * dv.y, dy.z * dv.z, dy.w * dv.w);
                if constexpr (SPLIT) {
                    // image [N][piece][C/8][HW][8 halves]: this thread's four channels are half an octet entry (8 bytes per piece); the block's
                    // pixel lanes are consecutive pixels, so a plane receives runs of ppb x 16 bytes
                    const float a0 = zz.x * sp_mul, a1 = zz.y * sp_mul, a2 = zz.z * sp_mul, a3 = zz.w * sp_mul;
                    const ue_fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(a0, a1), h23 = __builtin_amdgcn_cvt_pkrtz(a2, a3);
                    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                    const h2_t l01 = {(_Float16)__builtin_amdgcn_fmed3f((a0 - (float)h01[0]) * 2048.f, -65504.f, 65504.f),
                                      (_Float16)__builtin_amdgcn_fmed3f((a1 - (float)h01[1]) * 2048.f, -65504.f, 65504.f)};
                    const h2_t l23 = {(_Float16)__builtin_amdgcn_fmed3f((a2 - (float)h23[0]) * 2048.f, -65504.f, 65504.f),
                                      (_Float16)__builtin_amdgcn_fmed3f((a3 - (float)h23[1]) * 2048.f, -65504.f, 65504.f)};
                    uint2 hv, lv;
                    __builtin_memcpy(&hv.x, &h01, 4); __builtin_memcpy(&hv.y, &h23, 4);
                    __builtin_memcpy(&lv.x, &l01, 4); __builtin_memcpy(&lv.y, &l23, 4);
                    // staged [u][plane = piece * C/8 + octet][pixel lane] (16-byte entries, one pad entry per plane): written straight to the image
                    // a wave would touch 2 pixels x 16 bytes in each of 32 planes (2.6 TB/s); from LDS every thread stores one whole entry and 32
                    // (16) consecutive lanes one 512 (256)-byte run of a plane
                    char* sp = reinterpret_cast<char*>(red) + ((u * C4 + (c4 >> 1)) * (ppb + 1) + pl) * 16 + (c4 & 1) * 8;
                    *reinterpret_cast<uint2*>(sp) = hv;
                    *reinterpret_cast<uint2*>(sp + (C4 / 2) * (ppb + 1) * 16) = lv;
                    if (dz != nullptr) st4(dz + off, zz);
                } else {
                    st4(dz + off, zz);
                }
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(zz.x), fabsf(zz.y)), fmaxf(fabsf(zz.z), fabsf(zz.w))));
                accb.x += dy.x; accb.y += dy.y; accb.z += dy.z; accb.w += dy.w;
                const float nz = nraw[u] * strength;
                if (dd) {
                    accd.x += dy.x * (pre.x - bv.x - nz); accd.y += dy.y * (pre.y - bv.y - nz);
                    accd.z += dy.z * (pre.z - bv.z - nz); accd.w += dy.w * (pre.w - bv.w - nz);
                }
                if (dnoise || dstrength) {
                    float s = (dy.x + dy.y) + (dy.z + dy.w);
                    if (pow2) {
                        s = eg3d_row_group_sum(s, grp);
                        if ((threadIdx.x & (grp - 1)) == 0) {
                            if (dnoise) eg3d_acc(dnoise + (int64_t)n * dnoise_nstride + pix, s * strength);
                            accs += s * nraw[u];
                        }
                    } else {
                        if (dnoise) eg3d_acc(dnoise + (int64_t)n * dnoise_nstride + pix, s * strength);
                        accs += s * nraw[u];
                    }
                }
            }
            if constexpr (SPLIT) {
                lds_barrier();
                // item = (plane, pixel lane): EPI_BWD_THREADS = C4 * ppb of them per u
                const int plane = threadIdx.x / ppb, px = threadIdx.x - plane * ppb;
                const int piece = plane / (C4 / 2), ko = plane - piece * (C4 / 2);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pix = pix0 - pl + px + u * stride;
                    if (pix < HW) {
                        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(red) + ((u * C4 + plane) * (ppb + 1) + px) * 16);
                        reinterpret_cast<uint4*>(fin.simg)[((int64_t)(n * 2 + piece) * (C / 8) + ko) * HW + pix] = v;
                    }
                }
                lds_barrier();
                cur = nxt;
            }
        }
    }
    // block reduction over the PPB pixel lanes
    float* rb = red;                         // [ppb][C4][4]
    float* rd = red + ppb * C4 * 4;          // [ppb][C4][4]
    float* rz = rd + ppb * C4 * 4;           // [ppb][C4][4]  (FIN only)
    float* rs = rz + (FIN ? ppb * C4 * 4 : 0);    // [1]
    if (threadIdx.x == 0) rs[0] = 0.f;
    if (active) { st4(rb + (pl * C4 + c4) * 4, accb); st4(rd + (pl * C4 + c4) * 4, accd); }
    if (FIN && active) st4(rz + (pl * C4 + c4) * 4, accz);
    __syncthreads();
    if (dstrength && accs != 0.f) EG3D_LDS_ACC(rs, dstrength, accs);
    // pixel lanes -> one sum per channel: a tree over the pixel index (with 4 channels there are 1024 pixel lanes per block: summed by one
    // thread this tail took longer than the pass itself), skipped when no per-channel sum is asked for (a clamp-only pass)
    const bool want_cs = dbias != nullptr || dd != nullptr || (FIN && fin.ds != nullptr);
    if (want_cs) {
        for (int st = 1; st < ppb; st <<= 1) {
            if (active && (pl & (2 * st - 1)) == 0 && pl + st < ppb) {
                float* a0 = rb + (pl * C4 + c4) * 4;
                float* a1 = rd + (pl * C4 + c4) * 4;
                const float4 t = ld4(a0 + st * C4 * 4), u = ld4(a1 + st * C4 * 4), t0 = ld4(a0), u0 = ld4(a1);
                st4(a0, make_float4(t0.x + t.x, t0.y + t.y, t0.z + t.z, t0.w + t.w));
                st4(a1, make_float4(u0.x + u.x, u0.y + u.y, u0.z + u.z, u0.w + u.w));
                if (FIN) {
                    float* a2 = rz + (pl * C4 + c4) * 4;
                    const float4 v = ld4(a2 + st * C4 * 4), v0 = ld4(a2);
                    st4(a2, make_float4(v0.x + v.x, v0.y + v.y, v0.z + v.z, v0.w + v.w));
                }
            }
            __syncthreads();
        }
    }
    if (want_cs && threadIdx.x < C4) {
        const float4 sb = ld4(rb + c4 * 4), sd = ld4(rd + c4 * 4);
        if (dbias) {
            eg3d_acc(dbias + c + 0, sb.x); eg3d_acc(dbias + c + 1, sb.y);
            eg3d_acc(dbias + c + 2, sb.z); eg3d_acc(dbias + c + 3, sb.w);
        }
        if (dd) {     // dL/dd = sum dy * z,  z = (pre - bias - noise) / d
            float* q = dd + (int64_t)n * C + c;
            eg3d_acc(q + 0, sd.x / dv.x); eg3d_acc(q + 1, sd.y / dv.y);
            eg3d_acc(q + 2, sd.z / dv.z); eg3d_acc(q + 3, sd.w / dv.w);
        }
        if (FIN && fin.ds) {
            const float4 sz = ld4(rz + c4 * 4);
            float* q = fin.ds + (int64_t)n * C + c;
            eg3d_acc(q + 0, sz.x); eg3d_acc(q + 1, sz.y); eg3d_acc(q + 2, sz.z); eg3d_acc(q + 3, sz.w);
        }
    }
    __syncthreads();
    if (dstrength && threadIdx.x == 0 && rs[0] != 0.f) eg3d_acc(dstrength, rs[0]);
    if (dz_amax != nullptr) {                 // max|dz|: non-negative floats order like their bit patterns; one global atomic per block
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        unsigned* wmax = reinterpret_cast<unsigned*>(red);                 // the reduction scratch is free again
        __syncthreads();
        if (threadIdx.x == 0) wmax[0] = 0u;
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && amax < 3.0e38f) atomicMax(wmax, __float_as_uint(amax));
        __syncthreads();
        if (threadIdx.x == 0 && wmax[0] != 0u) atomicMax(reinterpret_cast<unsigned*>(dz_amax), wmax[0]);
    }
}

// finish of a split-K data-gradient: dx = z * s[n,c] (+ addend);  ds[n,c] += sum_px z * x      (grid = (blocks, N))
__global__ void __launch_bounds__(EPI_BWD_THREADS) dgrad_finish_kernel(const float* __restrict__ z, const float* __restrict__ x, const float* __restrict__ s,
                                                           const float* __restrict__ addend, float* __restrict__ dx, float* __restrict__ ds, int HW, int C4) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int n = blockIdx.y, C = C4 * 4;
    const int ppb = EPI_BWD_THREADS / C4 > 0 ? EPI_BWD_THREADS / C4 : 1;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const bool active = pl < ppb;
    const int c = c4 * 4;
    float4 sv = make_float4(1, 1, 1, 1), acc = make_float4(0, 0, 0, 0);
    if (active && s) sv = ld4(s + (int64_t)n * C + c);
    if (active) {
        constexpr int U = 4;                      // loads of U pixels in flight per trip (see epilogue_bwd_kernel)
        const int stride = gridDim.x * ppb;
        for (int pix0 = blockIdx.x * ppb + pl; pix0 < HW; pix0 += U * stride) {
            float4 zv[U], xv[U], av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = pix0 + u * stride;
                const int64_t off = ((int64_t)n * HW + (pix < HW ? pix : pix0)) * C + c;
                zv[u] = ld4(z + off);
                xv[u] = ds ? ld4(x + off) : make_float4(0, 0, 0, 0);
                av[u] = addend ? ld4(addend + off) : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pix = pix0 + u * stride;
                if (pix >= HW) break;
                const int64_t off = ((int64_t)n * HW + pix) * C + c;
                acc.x += zv[u].x * xv[u].x; acc.y += zv[u].y * xv[u].y; acc.z += zv[u].z * xv[u].z; acc.w += zv[u].w * xv[u].w;
                st4(dx + off, make_float4(zv[u].x * sv.x + av[u].x, zv[u].y * sv.y + av[u].y, zv[u].z * sv.z + av[u].z, zv[u].w * sv.w + av[u].w));
            }
        }
    }
    if (ds == nullptr) return;
    if (active) st4(red + (pl * C4 + c4) * 4, acc);
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 t = make_float4(0, 0, 0, 0);
        for (int q = 0; q < ppb; ++q) { float4 u = ld4(red + (q * C4 + c4) * 4); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        float* q = ds + (int64_t)n * C + c;
        eg3d_acc(q + 0, t.x); eg3d_acc(q + 1, t.y); eg3d_acc(q + 2, t.z); eg3d_acc(q + 3, t.w);
    }
}

__global__ void weight_sqsum_kernel(const float* __restrict__ w, float* __restrict__ wsq, int Co, int ntaps, int Ck) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * Ck) return;
    int o = i / Ck, k = i - o * Ck;
    float s = 0.f;
    for (int t = 0; t < ntaps; ++t) { float v = w[((int64_t)o * ntaps + t) * Ck + k]; s += v * v; }
    wsq[i] = s;
}

// One pass over a conv weight [O,I,T] (T = kh*kw taps) producing the three images the path keeps per layer: the forward operand
// wf[o][t*I + i], the data-gradient operand wa[i][t*O + o] and wsq[o][i] = sum_t w^2 (demodulation).  A block handles 32 x 32 (o,i)
// pairs through LDS so that all three are written in runs of 32 consecutive floats.  Replaces two permute-copies and a reduction per
// layer and step of the pivotal-tuning phase (weights change every step there).
constexpr int PK = 16;      // 16 x 16 (o, i) pairs per block: 1024 blocks for a 512 x 512 layer (32 x 32 left most CUs with a single, serial block)
__device__ __forceinline__ void pack_conv_weight_tile(float* sm, const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wa,
                                                      float* __restrict__ wsq, int O, int I, int T, const float* __restrict__ oscale, int Old, int bx, int by) {
    const int ld = PK * T + 1;                          // sm: [PK][PK*T + 1]
    const int o0 = by * PK, i0 = bx * PK;
    const int no = min(PK, O - o0), ni = min(PK, I - i0);
    const int tid = threadIdx.x, run = ni * T;
    for (int idx = tid; idx < no * run; idx += 256) {   // rows of w are contiguous in (i, t)
        const int ol = idx / run, r = idx - ol * run;
        sm[ol * ld + r] = w[((int64_t)(o0 + ol) * I + i0) * T + r] * (oscale ? oscale[o0 + ol] : 1.f);     // per-output-channel scale (folded BatchNorm)
    }
    __syncthreads();
    for (int idx = tid; idx < no * T * PK; idx += 256) {
        const int il = idx % PK, t = (idx / PK) % T, ol = idx / (PK * T);
        if (il < ni) wf[((int64_t)(o0 + ol) * T + t) * I + i0 + il] = sm[ol * ld + il * T + t];
    }
    if (wa != nullptr)
        for (int idx = tid; idx < ni * T * PK; idx += 256) {
            const int ol = idx % PK, t = (idx / PK) % T, il = idx / (PK * T);
            if (ol < no) wa[((int64_t)(i0 + il) * T + t) * Old + o0 + ol] = sm[ol * ld + il * T + t];       // Old >= O: rows padded by the caller
        }
    if (wsq != nullptr)
        for (int idx = tid; idx < no * PK; idx += 256) {
            const int il = idx % PK, ol = idx / PK;
            if (il < ni) {
                float a = 0.f;
                for (int t = 0; t < T; ++t) { const float v = sm[ol * ld + il * T + t]; a += v * v; }
                wsq[(int64_t)(o0 + ol) * I + i0 + il] = a;
            }
        }
}

__global__ void __launch_bounds__(256) pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wa,
                                                               float* __restrict__ wsq, int O, int I, int T, const float* __restrict__ oscale, int Old) {
    extern __shared__ float sm[];
    pack_conv_weight_tile(sm, w, wf, wa, wsq, O, I, T, oscale, Old, blockIdx.x, blockIdx.y);
}

// Every layer of a network in one launch (the weights of all layers change together, once per pivotal-tuning step): block -> (layer, tile)
// through a prefix table held in the kernel arguments.
struct PackBatch { eg3d_pack_item it[EG3D_PACK_BATCH_MAX]; int blk0[EG3D_PACK_BATCH_MAX + 1]; int n; };
__global__ void __launch_bounds__(256) pack_conv_weights_batched_kernel(const PackBatch b) {
    extern __shared__ float sm[];
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.blk0[l + 1]) ++l;
    const eg3d_pack_item& q = b.it[l];
    const int local = blockIdx.x - b.blk0[l], tiles_i = (q.I + PK - 1) / PK;
    pack_conv_weight_tile(sm, q.w, q.wf, q.wa, q.wsq, q.O, q.I, q.T, q.oscale, q.O_pad > 0 ? q.O_pad : q.O, local % tiles_i, local / tiles_i);
}

// Gradient of a per-output-channel scaled weight w' = w * a[o] from the packed weight-gradient image g[o][t*Ip + i] of the conv that used
// w':  dw[o][i][t] = g[o][t*Ip + i] * a[o]  (the parameter's own layout),  da[o] = sum_{i,t} g[o][t*Ip + i] * w[o][i][t].
// One block per output channel; the row goes through LDS so that both global streams are contiguous.
__global__ void __launch_bounds__(256) unpack_weight_grad_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ a,
                                                                 float* __restrict__ dw, float* __restrict__ da, int I, int Ip, int T) {
    extern __shared__ float row[];                      // [T][Ip]
    __shared__ float red[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int64_t gro = (int64_t)o * T * Ip, wro = (int64_t)o * I * T;
    for (int k = tid; k < T * Ip; k += 256) row[k] = g[gro + k];
    __syncthreads();
    const float ao = a ? a[o] : 1.f;
    float s = 0.f;
    for (int k = tid; k < I * T; k += 256) {
        const int i = k / T, t = k - i * T;
        const float gv = row[t * Ip + i];
        s = fmaf(gv, w[wro + k], s);
        dw[wro + k] = gv * ao;
    }
    if (da == nullptr) return;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) da[o] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Weight gradient of a demodulated modulated conv in the parameter's own layout, both paths in one pass (block = output channel):
//   dw[o][i][t] = g[o][t*I + i]  (the conv's packed weight-gradient image)  +  2 w[o][i][t] * dwsq[o][i],
//   dwsq[o][i] = sum_n dd[n,o] * (-1/2) d[n,o]^3 s[n,i]^2       (d = rsqrt(sum_i s^2 wsq + eps), wsq = sum_t w^2: networks_stylegan2.py:60-63)
// dd = gradient w.r.t. the demodulation coefficients (the activation-backward pass accumulates it); null: no demodulation term.
__global__ void __launch_bounds__(256) weight_grad_finish_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ s,
                                                                 const float* __restrict__ d, const float* __restrict__ dd, float* __restrict__ dw, int N,
                                                                 int O, int I, int T, int nslab, int64_t slab_stride) {
    extern __shared__ float row[];                      // [T][I + 1] packed row, then q[I] = 2 dwsq[o][.]
    const int o = blockIdx.x, tid = threadIdx.x, ld = I + 1;
    float* q = row + T * ld;
    const int64_t ro = (int64_t)o * T * I;
    for (int k = tid; k < T * I; k += 256) {
        const int t = k / I, i = k - t * I;
        float v = g[ro + k];
        for (int sl = 1; sl < nslab; ++sl) v += g[(int64_t)sl * slab_stride + ro + k];          // partial images, always in slab order
        row[t * ld + i] = v;
    }
    for (int i = tid; i < I; i += 256) {
        float acc = 0.f;
        if (dd != nullptr)
            for (int n = 0; n < N; ++n) {
                const float dv = d[(int64_t)n * O + o], sv = s[(int64_t)n * I + i];
                acc = fmaf(dd[(int64_t)n * O + o] * (-0.5f) * dv * dv * dv, sv * sv, acc);
            }
        q[i] = 2.f * acc;
    }
    __syncthreads();
    for (int k = tid; k < I * T; k += 256) {
        const int i = k / T, t = k - i * T;
        dw[ro + k] = fmaf(w[ro + k], q[i], row[t * ld + i]);
    }
}

// ... of every layer in one launch (pivotal tuning: 17 conv layers, block -> (layer, output channel) through a prefix table in the kernel
// arguments): the per-layer launches are 5 - 13 us each at 128 - 512 blocks.
struct WgfBatch { eg3d_wgf_item it[EG3D_WGF_BATCH_MAX]; int blk0[EG3D_WGF_BATCH_MAX + 1]; int n; };
__global__ void __launch_bounds__(256) weight_grad_finish_batched_kernel(const WgfBatch b) {
    extern __shared__ float row[];
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.blk0[l + 1]) ++l;
    const eg3d_wgf_item& q = b.it[l];
    const int o = blockIdx.x - b.blk0[l], tid = threadIdx.x, I = q.I, T = q.T, O = q.O, ld = I + 1;
    float* qq = row + T * ld;
    const int64_t ro = (int64_t)o * T * I;
    for (int k = tid; k < T * I; k += 256) {
        const int t = k / I, i = k - t * I;
        float v = q.g[ro + k];
        for (int sl = 1; sl < q.nslab; ++sl) v += q.g[(int64_t)sl * q.slab_stride + ro + k];
        row[t * ld + i] = v;
    }
    for (int i = tid; i < I; i += 256) {
        float acc = 0.f;
        if (q.dd != nullptr)
            for (int n = 0; n < q.N; ++n) {
                const float dv = q.d[(int64_t)n * O + o], sv = q.s[(int64_t)n * I + i];
                acc = fmaf(q.dd[(int64_t)n * O + o] * (-0.5f) * dv * dv * dv, sv * sv, acc);
            }
        qq[i] = 2.f * acc;
    }
    __syncthreads();
    for (int k = tid; k < I * T; k += 256) {
        const int i = k / T, t = k - i * T;
        q.dw[ro + k] = fmaf(q.w[ro + k], qq[i], row[t * ld + i]);
    }
}

// one wave per (n,o)
__global__ void __launch_bounds__(256) demod_fwd_kernel(const float* __restrict__ s, const float* __restrict__ wsq, float* __restrict__ d, int N, int Co, int Ck) {
    int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= N * Co) return;
    int n = wid / Co, o = wid - n * Co;
    float acc = 0.f;
    for (int k = lane; k < Ck; k += 64) { float sv = s[(int64_t)n * Ck + k]; acc += sv * sv * wsq[(int64_t)o * Ck + k]; }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) d[wid] = 1.0f / sqrtf(acc + 1e-8f);
}

// ds[n,k] += -s[n,k] * sum_o dd[n,o] d[n,o]^3 wsq[o,k].  Block = 64 k-columns x 4 o-slices (coalesced rows of wsq), grid.y = n,
// grid.z splits o further; partial sums meet in LDS, one atomic per (block, k).
__global__ void __launch_bounds__(256) demod_bwd_kernel(const float* __restrict__ s, const float* __restrict__ wsq, const float* __restrict__ d,
                                                        const float* __restrict__ dd, float* __restrict__ ds, int N, int Co, int Ck, int osplit) {
    __shared__ float part[4][64];
    const int n = blockIdx.y;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int per = (Co + osplit - 1) / osplit;
    const int o_beg = blockIdx.z * per, o_end = min(Co, o_beg + per);
    float acc = 0.f;
    if (k < Ck) {
        for (int o = o_beg + sl; o < o_end; o += 4) {
            float dv = d[(int64_t)n * Co + o];
            acc = fmaf(dd[(int64_t)n * Co + o] * dv * dv * dv, wsq[(int64_t)o * Ck + k], acc);
        }
    }
    part[sl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sl == 0 && k < Ck) {
        float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        eg3d_acc(ds + (int64_t)n * Ck + k, -s[(int64_t)n * Ck + k] * t);
    }
}

// thread per (o,k): dwsq[o,k] += sum_n dd[n,o] * (-0.5 d^3 s[n,k]^2)
__global__ void __launch_bounds__(256) demod_bwd_wsq_kernel(const float* __restrict__ s, const float* __restrict__ d, const float* __restrict__ dd,
                                                            float* __restrict__ dwsq, int N, int Co, int Ck) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * Ck) return;
    int o = i / Ck, k = i - o * Ck;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        float dv = d[(int64_t)n * Co + o], sv = s[(int64_t)n * Ck + k];
        acc += dd[(int64_t)n * Co + o] * (-0.5f) * dv * dv * dv * sv * sv;
    }
    dwsq[i] += acc;
}

}  // namespace

extern "C" int eg3d_modconv_epilogue_fwd(const float* z, float* out, int N, int H, int W, int C, int Hz, int Wz, const float* fir, int fh,
                                         int fw, int pad0, float fir_gain, const float* d, const float* noise, int64_t noise_nstride,
                                         const float* noise_strength, const float* bias, int act, float alpha, float gain, float clamp,
                                         float* out_amax, void* stream) {
    if (!z || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return EG3D_ERR_INVALID;
    if (C % 4) return EG3D_ERR_UNSUPPORTED;
    if (fir && (fh < 1 || fw < 1 || fh * fw > 64)) return EG3D_ERR_UNSUPPORTED;
    if (fir && (H != Hz + 2 * pad0 - fh + 1 || W != Wz + 2 * pad0 - fw + 1)) return EG3D_ERR_INVALID;
    if (!fir && (Hz != H || Wz != W)) return EG3D_ERR_INVALID;
    if (noise && !noise_strength) return EG3D_ERR_INVALID;
    const bool pwl = eg3d_act_is_pwl(act);
    const float slope = eg3d_act_pwl_slope(act, alpha);
    if (fir != nullptr && fh == 4 && fw == 4 && !(H & 1) && !(W & 1)) {
        const int64_t total4 = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
        const int blocks4 = (int)std::min<int64_t>(eg3d_cdiv(total4, 256), 256 * 16);
        if (pwl) hipLaunchKernelGGL(epilogue_fwd_fir44_kernel<true>, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, z, out, N, H, W, C / 4, Hz, Wz, fir, pad0,
                                    fir_gain, d, noise, noise_nstride, noise_strength, bias, act, slope, gain, clamp, out_amax);
        else hipLaunchKernelGGL(epilogue_fwd_fir44_kernel<false>, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, z, out, N, H, W, C / 4, Hz, Wz, fir, pad0,
                                fir_gain, d, noise, noise_nstride, noise_strength, bias, act, alpha, gain, clamp, out_amax);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
    const int64_t total = (int64_t)N * H * W * (C / 4);
    int blocks = (int)std::min<int64_t>(eg3d_cdiv(total, 256), 256 * 16);
    if (pwl) hipLaunchKernelGGL(epilogue_fwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, out, N, H, W, C / 4, Hz, Wz, fir, fh, fw, pad0,
                                fir_gain, d, noise, noise_nstride, noise_strength, bias, act, slope, gain, clamp, out_amax);
    else hipLaunchKernelGGL(epilogue_fwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, out, N, H, W, C / 4, Hz, Wz, fir, fh, fw, pad0,
                            fir_gain, d, noise, noise_nstride, noise_strength, bias, act, alpha, gain, clamp, out_amax);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_modconv_epilogue_bwd(const float* dout, const float* out, float* dz, int N, int H, int W, int C, const float* d,
                                         const float* noise, int64_t noise_nstride, const float* noise_strength, const float* bias, int act,
                                         float alpha, float gain, float clamp, float* dbias, float* dd, float* dnoise, int64_t dnoise_nstride,
                                         float* dstrength, float* dz_amax, void* stream) {
    if (!dout || !out || !dz || N <= 0 || H <= 0 || W <= 0 || C <= 0) return EG3D_ERR_INVALID;
    if (C % 4 || C / 4 > 256) return EG3D_ERR_UNSUPPORTED;
    if (dd && act != EG3D_ACT_LINEAR && act != EG3D_ACT_LRELU) return EG3D_ERR_UNSUPPORTED;   // needs an invertible activation
    if (dd && !d) return EG3D_ERR_INVALID;
    if ((noise || dnoise || dstrength) && !noise_strength) return EG3D_ERR_INVALID;
    if ((dnoise || dstrength) && !noise) return EG3D_ERR_INVALID;
    const int C4 = C / 4;
    const int ppb = std::max(EPI_BWD_THREADS / C4, 1);
    // one atomic per (block, channel) lands on the same N*C addresses: keep the block count near the CU count
    // >= 4 pixels per thread (one unrolled trip; 8 was 5 us slower on the 8^2..32^2 layers, equal above) so that the per-block channel atomics stay a small fraction of the work
    const int cap = epi_bwd_cap();
    int bx = std::max(1, std::min(eg3d_cdiv((int64_t)H * W, ppb * 4), std::max(1, cap / N)));
    size_t smem = (size_t)(ppb * C4 * 8 + 4) * sizeof(float);
    const FinArgs nofin = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    EG3D_DET_SCOPE(det, stream);
    EG3D_DET_BIND(det, dbias, C); EG3D_DET_BIND(det, dd, (int64_t)N * C); EG3D_DET_BIND(det, dnoise, (int64_t)(N - 1) * dnoise_nstride + (int64_t)H * W);
    EG3D_DET_BIND(det, dstrength, 1);
    EG3D_DET_COMMIT(det);
    if (eg3d_act_is_pwl(act))
        hipLaunchKernelGGL((epilogue_bwd_kernel<true, false>), dim3(bx, N), dim3(EPI_BWD_THREADS), smem, (hipStream_t)stream, nofin, dout, out, dz, H, W, C4, d, noise, noise_nstride,
                           noise_strength, bias, act, eg3d_act_pwl_slope(act, alpha), gain, clamp, dbias, dd, dnoise, dnoise_nstride, dstrength, dz_amax);
    else
        hipLaunchKernelGGL((epilogue_bwd_kernel<false, false>), dim3(bx, N), dim3(EPI_BWD_THREADS), smem, (hipStream_t)stream, nofin, dout, out, dz, H, W, C4, d, noise, noise_nstride,
                           noise_strength, bias, act, alpha, gain, clamp, dbias, dd, dnoise, dnoise_nstride, dstrength, dz_amax);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_dgrad_finish_act(const float* z, const float* x, const float* s, const float* addend, float* dz, float* ds, int N, int H, int W, int C,
                                     const eg3d_act_bwd* ab, float* dz_amax, void* stream) {
    if (!z || !x || !dz || !ab || N <= 0 || H <= 0 || W <= 0 || C <= 0) return EG3D_ERR_INVALID;
    if (C % 4 || C / 4 > 256) return EG3D_ERR_UNSUPPORTED;
    if (ab->act != EG3D_ACT_LINEAR && ab->act != EG3D_ACT_LRELU) return EG3D_ERR_UNSUPPORTED;
    if ((ab->dd && !ab->d) || ((ab->noise || ab->dnoise || ab->dstrength) && !ab->noise_strength) || ((ab->dnoise || ab->dstrength) && !ab->noise)) return EG3D_ERR_INVALID;
    const int C4 = C / 4, ppb = std::max(EPI_BWD_THREADS / C4, 1);
    int bx = std::max(1, std::min(eg3d_cdiv((int64_t)H * W, ppb * 4), std::max(1, epi_bwd_cap() / N)));
    const size_t smem = (size_t)(ppb * C4 * 12 + 4) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    auto kern = epilogue_bwd_kernel<true, true>;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem, attr_done)) return e;
    const FinArgs fin = {z, s, addend, ds, nullptr, nullptr};
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, ds, (int64_t)N * C); EG3D_DET_BIND_ACT(det, *ab, N, C, (int64_t)H * W); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(kern, dim3(bx, N), dim3(EPI_BWD_THREADS), smem, (hipStream_t)stream, fin, nullptr, x, dz, H, W, C4, ab->d, ab->noise, ab->noise_nstride,
                       ab->noise_strength, ab->bias, ab->act, eg3d_act_pwl_slope(ab->act, ab->alpha), ab->gain, ab->clamp, ab->dbias, ab->dd, ab->dnoise,
                       ab->dnoise_nstride, ab->dstrength, dz_amax);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_torgb_dgrad_act(const float* dy4, const float* wa4, const float* x, const float* s, const float* addend, float* dz, float* ds, int N, int H,
                                    int W, int C, const eg3d_act_bwd* ab, float* dz_amax, void* stream) {
    if (!dy4 || !wa4 || !x || !dz || !ab || N <= 0 || H <= 0 || W <= 0 || C <= 0) return EG3D_ERR_INVALID;
    if (C % 4 || C / 4 > 256 || (reinterpret_cast<uintptr_t>(dy4) & 15) || (reinterpret_cast<uintptr_t>(wa4) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (ab->act != EG3D_ACT_LINEAR && ab->act != EG3D_ACT_LRELU) return EG3D_ERR_UNSUPPORTED;
    if ((ab->dd && !ab->d) || ((ab->noise || ab->dnoise || ab->dstrength) && !ab->noise_strength) || ((ab->dnoise || ab->dstrength) && !ab->noise)) return EG3D_ERR_INVALID;
    const int C4 = C / 4, ppb = std::max(EPI_BWD_THREADS / C4, 1);
    int bx = std::max(1, std::min(eg3d_cdiv((int64_t)H * W, ppb * 4), std::max(1, epi_bwd_cap() / N)));
    const size_t smem = (size_t)(ppb * C4 * 12 + 4) * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    auto kern = epilogue_bwd_kernel<true, true>;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem, attr_done)) return e;
    const FinArgs fin = {nullptr, s, addend, ds, dy4, wa4, nullptr, nullptr, nullptr, nullptr};
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, ds, (int64_t)N * C); EG3D_DET_BIND_ACT(det, *ab, N, C, (int64_t)H * W); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(kern, dim3(bx, N), dim3(EPI_BWD_THREADS), smem, (hipStream_t)stream, fin, nullptr, x, dz, H, W, C4, ab->d, ab->noise, ab->noise_nstride,
                       ab->noise_strength, ab->bias, ab->act, eg3d_act_pwl_slope(ab->act, ab->alpha), ab->gain, ab->clamp, ab->dbias, ab->dd, ab->dnoise,
                       ab->dnoise_nstride, ab->dstrength, dz_amax);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_torgb_dgrad_act_split(const float* dy4, const float* wa4, const float* x, const float* s, const float* addend, float* dz, float* ds, int N,
                                          int H, int W, int C, const eg3d_act_bwd* ab, const float* dy_amax, const float* addend_amax, void* split_image,
                                          float* split_scale_out, void* stream) {
    if (!dy4 || !wa4 || !x || !ab || !dy_amax || !split_image || !split_scale_out || (addend && !addend_amax) || N <= 0 || H <= 0 || W <= 0 || C <= 0)
        return EG3D_ERR_INVALID;
    if (C % 8 || C / 4 > 256 || (reinterpret_cast<uintptr_t>(dy4) & 15) || (reinterpret_cast<uintptr_t>(wa4) & 15) || (reinterpret_cast<uintptr_t>(split_image) & 15))
        return EG3D_ERR_UNSUPPORTED;
    if (ab->act != EG3D_ACT_LINEAR && ab->act != EG3D_ACT_LRELU) return EG3D_ERR_UNSUPPORTED;
    if ((ab->dd && !ab->d) || ((ab->noise || ab->dnoise || ab->dstrength) && !ab->noise_strength) || ((ab->dnoise || ab->dstrength) && !ab->noise)) return EG3D_ERR_INVALID;
    if ((int64_t)N * C > (1 << 20)) return EG3D_ERR_TOO_LARGE;
    const int C4 = C / 4, ppb = std::max(EPI_BWD_THREADS / C4, 1);
    int bx = std::max(1, std::min(eg3d_cdiv((int64_t)H * W, ppb * 4), std::max(1, epi_bwd_cap() / N)));
    if (EPI_BWD_THREADS % C4) return EG3D_ERR_UNSUPPORTED;                  // every thread takes part in the staging barriers
    const size_t smem = std::max((size_t)(ppb * C4 * 12 + 4) * sizeof(float), (size_t)4 * C4 * (ppb + 1) * 16);      // reductions | staged pieces of 4 pixels per thread
    static std::atomic<uint64_t> attr_done{0};
    auto kern = epilogue_bwd_kernel<true, true, true>;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem, attr_done)) return e;
    const FinArgs fin = {nullptr, s, addend, ds, dy4, wa4, reinterpret_cast<uint2*>(split_image), dy_amax, addend ? addend_amax : nullptr, split_scale_out};
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, ds, (int64_t)N * C); EG3D_DET_BIND_ACT(det, *ab, N, C, (int64_t)H * W); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(kern, dim3(bx, N), dim3(EPI_BWD_THREADS), smem, (hipStream_t)stream, fin, nullptr, x, dz, H, W, C4, ab->d, ab->noise, ab->noise_nstride,
                       ab->noise_strength, ab->bias, ab->act, eg3d_act_pwl_slope(ab->act, ab->alpha), ab->gain, ab->clamp, ab->dbias, ab->dd, ab->dnoise,
                       ab->dnoise_nstride, ab->dstrength, nullptr);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_dgrad_finish(const float* z, const float* x, const float* s, const float* addend, float* dx, float* ds, int N, int H, int W, int C,
                                 void* stream) {
    if (!z || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (ds && !x)) return EG3D_ERR_INVALID;
    if (C % 4 || C / 4 > 256) return EG3D_ERR_UNSUPPORTED;
    const int C4 = C / 4, ppb = std::max(EPI_BWD_THREADS / C4, 1);
    int bx = std::max(1, std::min(eg3d_cdiv((int64_t)H * W, ppb * 8), std::max(1, epi_bwd_cap() / N)));
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, ds, (int64_t)N * C); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(dgrad_finish_kernel, dim3(bx, N), dim3(EPI_BWD_THREADS), (size_t)ppb * C4 * 4 * sizeof(float), (hipStream_t)stream, z, x, s, addend, dx, ds,
                       H * W, C4);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_weight_sqsum(const float* w, float* wsq, int Co, int ntaps, int Ck, void* stream) {
    if (!w || !wsq || Co <= 0 || ntaps <= 0 || Ck <= 0) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(weight_sqsum_kernel, dim3(eg3d_cdiv((int64_t)Co * Ck, 256)), dim3(256), 0, (hipStream_t)stream, w, wsq, Co, ntaps, Ck);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_weight_grad_finish_batched(const eg3d_wgf_item* items, int n, void* stream) {
    if (!items || n < 1 || n > EG3D_WGF_BATCH_MAX) return EG3D_ERR_INVALID;
    WgfBatch b;
    int blocks = 0;
    size_t smem = 0;
    for (int l = 0; l < n; ++l) {
        const eg3d_wgf_item& q = items[l];
        if (!q.g || !q.w || !q.dw || q.N <= 0 || q.O <= 0 || q.I <= 0 || q.T <= 0 || q.T > 64 || (q.dd && (!q.s || !q.d)) || q.nslab < 1 ||
            (q.nslab > 1 && q.slab_stride < (int64_t)q.O * q.T * q.I)) return EG3D_ERR_INVALID;
        b.it[l] = q;
        b.blk0[l] = blocks;
        blocks += q.O;
        smem = std::max(smem, ((size_t)q.T * (q.I + 1) + q.I) * sizeof(float));
    }
    b.blk0[n] = blocks;
    b.n = n;
    if (smem > 64 * 1024) return EG3D_ERR_UNSUPPORTED;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(weight_grad_finish_batched_kernel), 64 * 1024, attr_done)) return e;
    hipLaunchKernelGGL(weight_grad_finish_batched_kernel, dim3(blocks), dim3(256), smem, (hipStream_t)stream, b);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

static int pack_conv_weight_impl(const float* w, const float* oscale, float* wf, float* wa, float* wsq, int O, int I, int T, void* stream, int Old = 0);

extern "C" int eg3d_pack_conv_weight(const float* w, float* wf, float* wa, float* wsq, int O, int I, int T, void* stream) {
    if (!wa) return EG3D_ERR_INVALID;
    return pack_conv_weight_impl(w, nullptr, wf, wa, wsq, O, I, T, stream);
}

extern "C" int eg3d_pack_conv_weight_scaled(const float* w, const float* oscale, float* wf, float* wa, int O, int I, int T, void* stream) {
    return pack_conv_weight_impl(w, oscale, wf, wa, nullptr, O, I, T, stream);
}

extern "C" int eg3d_pack_conv_weights_batched(const eg3d_pack_item* items, int n, void* stream) {
    if (!items || n < 1 || n > EG3D_PACK_BATCH_MAX) return EG3D_ERR_INVALID;
    PackBatch b;
    int maxT = 1, blocks = 0;
    for (int l = 0; l < n; ++l) {
        const eg3d_pack_item& q = items[l];
        if (!q.w || !q.wf || q.O <= 0 || q.I <= 0 || q.T <= 0 || q.T > 64 || (q.O_pad != 0 && q.O_pad < q.O)) return EG3D_ERR_INVALID;
        b.it[l] = q;
        b.blk0[l] = blocks;
        blocks += eg3d_cdiv(q.I, PK) * eg3d_cdiv(q.O, PK);
        maxT = std::max(maxT, q.T);
    }
    b.blk0[n] = blocks;
    b.n = n;
    const size_t smem = (size_t)PK * (PK * maxT + 1) * sizeof(float);
    if (smem > 64 * 1024) return EG3D_ERR_UNSUPPORTED;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(pack_conv_weights_batched_kernel), 64 * 1024, attr_done)) return e;
    hipLaunchKernelGGL(pack_conv_weights_batched_kernel, dim3(blocks), dim3(256), smem, (hipStream_t)stream, b);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_pack_conv_weight_padded(const float* w, float* wf, float* wa, float* wsq, int O, int I, int T, int O_pad, void* stream) {
    return pack_conv_weight_impl(w, nullptr, wf, wa, wsq, O, I, T, stream, O_pad);
}

extern "C" int eg3d_unpack_weight_grad(const float* g, const float* w, const float* oscale, float* dw, float* doscale, int O, int I, int Ip, int T, void* stream) {
    if (!g || !w || !dw || O <= 0 || I <= 0 || Ip < I || T <= 0 || T > 64) return EG3D_ERR_INVALID;
    const size_t smem = (size_t)T * Ip * sizeof(float);
    if (smem > 64 * 1024) return EG3D_ERR_UNSUPPORTED;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(unpack_weight_grad_kernel), 64 * 1024, attr_done)) return e;
    hipLaunchKernelGGL(unpack_weight_grad_kernel, dim3(O), dim3(256), smem, (hipStream_t)stream, g, w, oscale, dw, doscale, I, Ip, T);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_weight_grad_finish(const float* g, const float* w, const float* s, const float* d, const float* dd, float* dw, int N, int O, int I,
                                      int T, void* stream) {
    return eg3d_weight_grad_finish_slabs(g, 1, 0, w, s, d, dd, dw, N, O, I, T, stream);
}

extern "C" int eg3d_weight_grad_finish_slabs(const float* g, int nslab, int64_t slab_stride, const float* w, const float* s, const float* d, const float* dd,
                                            float* dw, int N, int O, int I, int T, void* stream) {
    if (!g || !w || !dw || N <= 0 || O <= 0 || I <= 0 || T <= 0 || T > 64 || (dd && (!s || !d)) || nslab < 1 || (nslab > 1 && slab_stride < (int64_t)O * T * I)) return EG3D_ERR_INVALID;
    const size_t smem = ((size_t)T * (I + 1) + I) * sizeof(float);
    if (smem > 64 * 1024) return EG3D_ERR_UNSUPPORTED;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(weight_grad_finish_kernel), 64 * 1024, attr_done)) return e;
    hipLaunchKernelGGL(weight_grad_finish_kernel, dim3(O), dim3(256), smem, (hipStream_t)stream, g, w, s, d, dd, dw, N, O, I, T, nslab, slab_stride);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

static int pack_conv_weight_impl(const float* w, const float* oscale, float* wf, float* wa, float* wsq, int O, int I, int T, void* stream, int Old) {
    if (!w || !wf || O <= 0 || I <= 0 || T <= 0 || T > 64 || (Old != 0 && Old < O)) return EG3D_ERR_INVALID;
    if (Old == 0) Old = O;
    const size_t smem = (size_t)PK * (PK * T + 1) * sizeof(float);
    if (smem > 64 * 1024) return EG3D_ERR_UNSUPPORTED;
    static std::atomic<uint64_t> attr_done{0};
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(pack_conv_weight_kernel), 64 * 1024, attr_done)) return e;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(eg3d_cdiv(I, PK), eg3d_cdiv(O, PK)), dim3(256), smem, (hipStream_t)stream, w, wf, wa, wsq, O, I, T, oscale, Old);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_demod_fwd(const float* s, const float* wsq, float* d, int N, int Co, int Ck, void* stream) {
    if (!s || !wsq || !d || N <= 0 || Co <= 0 || Ck <= 0) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(demod_fwd_kernel, dim3(eg3d_cdiv((int64_t)N * Co * 64, 256)), dim3(256), 0, (hipStream_t)stream, s, wsq, d, N, Co, Ck);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_demod_bwd(const float* s, const float* wsq, const float* d, const float* dd, float* ds, float* dwsq, int N, int Co, int Ck,
                              void* stream) {
    if (!s || !wsq || !d || !dd || N <= 0 || Co <= 0 || Ck <= 0) return EG3D_ERR_INVALID;
    if (ds) {
        const int osplit = std::max(1, std::min(Co / 16, 16));
        EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, ds, (int64_t)N * Ck); EG3D_DET_COMMIT(det);
        hipLaunchKernelGGL(demod_bwd_kernel, dim3(eg3d_cdiv(Ck, 64), N, osplit), dim3(256), 0, (hipStream_t)stream, s, wsq, d, dd, ds, N, Co, Ck, osplit);
        EG3D_DET_END(det);
    }
    if (dwsq) hipLaunchKernelGGL(demod_bwd_wsq_kernel, dim3(eg3d_cdiv((int64_t)Co * Ck, 256)), dim3(256), 0, (hipStream_t)stream, s, d, dd, dwsq, N, Co, Ck);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}


// The up layers' epilogue for a separable 4-tap FIR (host taps k[4]; 2-D filter = outer(k, k)), C % 64 == 0, piecewise-linear activation.
// split_image / split_in_scale / split_scale_out (all or none; clamp >= 0 required): also write the consumer layer's operand image
// split(out * split_in_scale[n,c]) in the layout of eg3d_split_activation (eg3d_split_activation_bytes(N, H, W, C) bytes).
extern "C" int eg3d_upconv_epilogue_fwd(const float* z, float* out, int N, int H, int W, int C, int Hz, int Wz, const float* k4, int pad0, float fir_gain,
                                        const float* d, const float* noise, int64_t noise_nstride, const float* noise_strength, const float* bias, int act,
                                        float alpha, float gain, float clamp, float* out_amax, const float* split_in_scale, void* split_image,
                                        float* split_scale_out, void* stream) {
    if (!z || !out || !k4 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || Hz <= 0 || Wz <= 0) return EG3D_ERR_INVALID;
    if (C % UE_CH || !eg3d_act_is_pwl(act)) return EG3D_ERR_UNSUPPORTED;
    if (noise && !noise_strength) return EG3D_ERR_INVALID;
    const bool split = split_image != nullptr;
    if (split && (!split_in_scale || !split_scale_out || !(clamp >= 0.f))) return EG3D_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(z) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (split && (reinterpret_cast<uintptr_t>(split_image) & 15))) return EG3D_ERR_UNSUPPORTED;
    const dim3 grid(N * eg3d_cdiv(H, UE_TH) * eg3d_cdiv(W, UE_TW), C / UE_CH);
    const float slope = eg3d_act_pwl_slope(act, alpha);
    hipStream_t st = (hipStream_t)stream;
    if (split)
        hipLaunchKernelGGL(upconv_epilogue_kernel<true>, grid, dim3(256), 0, st, z, out, N, H, W, C, Hz, Wz, k4[0], k4[1], k4[2], k4[3], pad0, fir_gain, d, noise,
                           noise_nstride, noise_strength, bias, slope, gain, clamp, out_amax, split_in_scale, reinterpret_cast<_Float16*>(split_image), split_scale_out);
    else
        hipLaunchKernelGGL(upconv_epilogue_kernel<false>, grid, dim3(256), 0, st, z, out, N, H, W, C, Hz, Wz, k4[0], k4[1], k4[2], k4[3], pad0, fir_gain, d, noise,
                           noise_nstride, noise_strength, bias, slope, gain, clamp, out_amax, nullptr, nullptr, nullptr);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
