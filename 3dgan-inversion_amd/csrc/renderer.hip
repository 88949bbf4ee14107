// Fused tri-plane volume renderer for gfx950 (forward + backward), one (ray, sample) pair per thread.
//
// Replaces, per ray and without materialising anything but the per-ray outputs (the reference writes ~5 GB of
// intermediates per pass):
//   RaySampler.forward                       training/volumetric_rendering/ray_sampler.py:24-73
//   ImportanceRenderer.forward               renderer.py:143-195   (fixed / per-ray limits)
//     sample_stratified :224-247, sample_from_planes + F.grid_sample(bilinear, zeros, align_corners=False) :39-66,
//     OSGDecoder.forward triplane.py:124-136, MipRayMarcher2 ray_marcher.py:25-57,
//     sample_importance/sample_pdf :249-308, unify_samples :212-222
//
// Thread mapping: a block holds RPB rays x D threads (D = max(coarse, fine) samples, 48 at the FFHQ config -> 4 rays
// per 192-thread block).  Thread (r,s) evaluates coarse sample s and fine sample s of ray r: 12 texel gathers of 128 B
// (planes are NHWC so the 32 features of a texel are one cache line) + the 32->64->33 MLP in registers; decoder weights
// are wave-uniform and come through the scalar cache as SGPR operands of v_fmac (fp32 VALU rate == fp32 MFMA rate on
// gfx950, so the MLP stays on the vector pipe).  Everything that couples the samples of a ray (transmittance scan,
// importance CDF, merge of the sorted coarse list with the unsorted fine list, compositing) goes through a few hundred
// bytes of LDS per ray.  The composite colour uses  sum_i w_i (c_i + c_{i+1})/2 = sum_j c_j (w_{j-1} + w_j)/2  so the
// 2 x 32 colours of a thread never leave its registers.
#include "common.h"
#include "det.h"
#include "render_common.h"

namespace {

constexpr int FC = 32;     // features per plane
constexpr int HD = 64;     // decoder hidden width
constexpr int CO = 32;     // decoder colour outputs
#ifndef RK_MAXT
#define RK_MAXT 192
#endif
constexpr int MAXT = RK_MAXT;  // threads per block

// Decoder non-linearities: 64 softplus + 32 sigmoid per sample.  The libm log1pf/expf expand to ~150 instructions each (more than
// twice the MLP's FMAs); the hardware exp2/log2 forms below are ~10 instructions, absolute error < 2e-7 on the result
// (softplus(x) = max(x,0) + log(1 + exp(-|x|)) keeps the argument of log in (1,2]).  The ray marcher keeps the libm forms.
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_acc(float x) { return x > 20.f ? x : log1pf(expf(x)); }

struct PlaneUV { float u, v; };

__device__ __forceinline__ void plane_uv(int pl, float x, float y, float z, float& u, float& v) {
    // renderer.py:23-53: plane 0 -> (x,y), plane 1 -> (x,z), plane 2 -> (z,x)
    if (pl == 0) { u = x; v = y; } else if (pl == 1) { u = x; v = z; } else { u = z; v = x; }
}

// bilinear gather of the 3 planes at one point, mean over planes.  pn = planes + n*Hp*Wp*ldp.
__device__ __forceinline__ void gather_feats(const float* __restrict__ pn, int Hp, int Wp, int ldp, float cs, float x, float y, float z,
                                             float (&f)[FC]) {
#pragma unroll
    for (int c = 0; c < FC; ++c) f[c] = 0.f;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        float u, v;
        plane_uv(pl, x * cs, y * cs, z * cs, u, v);
        float ix = ((u + 1.f) * Wp - 1.f) * 0.5f, iy = ((v + 1.f) * Hp - 1.f) * 0.5f;
        float fx0 = floorf(ix), fy0 = floorf(iy);
        int x0 = (int)fx0, y0 = (int)fy0;
        float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
        const float wts[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int xx = x0 + (q & 1), yy = y0 + (q >> 1);
            if ((unsigned)xx < (unsigned)Wp && (unsigned)yy < (unsigned)Hp) {
                const float4* t = reinterpret_cast<const float4*>(pn + ((int64_t)yy * Wp + xx) * ldp + pl * FC);
                float w = wts[q];
#pragma unroll
                for (int c4 = 0; c4 < FC / 4; ++c4) {
                    float4 tv = t[c4];
                    f[c4 * 4 + 0] = fmaf(w, tv.x, f[c4 * 4 + 0]); f[c4 * 4 + 1] = fmaf(w, tv.y, f[c4 * 4 + 1]);
                    f[c4 * 4 + 2] = fmaf(w, tv.z, f[c4 * 4 + 2]); f[c4 * 4 + 3] = fmaf(w, tv.w, f[c4 * 4 + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < FC; ++c) f[c] = f[c] * (1.f / 3.f);
}

// 32 -> 64 (softplus) -> 33; w1t is [HD][1+CO] (transposed so that a hidden unit's fan-out is contiguous)
__device__ __forceinline__ void mlp_fwd(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1t,
                                        const float* __restrict__ b1, const float (&f)[FC], float (&out)[1 + CO]) {
#pragma unroll
    for (int k = 0; k < 1 + CO; ++k) out[k] = b1[k];
#pragma unroll 2
    for (int j = 0; j < HD; ++j) {
        const float* wj = w0 + j * FC;
        float pre = b0[j];
#pragma unroll
        for (int c = 0; c < FC; ++c) pre = fmaf(wj[c], f[c], pre);
        float h = softplusf_(pre);
        const float* vj = w1t + j * (1 + CO);
#pragma unroll
        for (int k = 0; k < 1 + CO; ++k) out[k] = fmaf(vj[k], h, out[k]);
    }
}

__device__ __forceinline__ float lin_depth(int i, int D, float start, float end) {
    // torch.linspace: symmetric evaluation from both ends
    float step = (end - start) / (float)(D - 1);
    return i < D / 2 ? start + step * (float)i : end - step * (float)(D - 1 - i);
}

__device__ __forceinline__ float coarse_depth(const eg3d_render_params& p, int64_t ray, int s, float u) {
    const int D = p.Dc;
    if (p.ray_limits != nullptr) {        // per-ray limits: math_utils.linspace (start + i/(D-1) * (end-start))
        float rs = p.ray_limits[ray * 2], re = p.ray_limits[ray * 2 + 1];
        float t = rs + ((float)s / (float)(D - 1)) * (re - rs);
        return t + u * ((re - rs) / (float)(D - 1));
    }
    if (p.disparity) {
        float t = lin_depth(s, D, 0.f, 1.f) + u * (1.f / (float)(D - 1));
        return 1.f / (1.f / p.ray_start * (1.f - t) + 1.f / p.ray_end * t);
    }
    return lin_depth(s, D, p.ray_start, p.ray_end) + u * ((p.ray_end - p.ray_start) / (float)(D - 1));
}

struct RayLds {          // per-ray LDS scratch (floats), laid out by the kernels below
    float* dc; float* sc; float* df; float* sf;   // coarse / fine depths and densities          [D] each
    float* sd; float* ss;                          // merged, depth-sorted depths and densities   [2D]
    float* w;                                      // interval weights                            [2D]
    float* q;                                      // 1 - alpha + 1e-10                           [2D]
    float* t;                                      // scratch                                     [2D]
    float* misc;                                   // [8]
};
constexpr int RAY_LDS_FLOATS(int D) { return 4 * D + 5 * 2 * D + 8; }

__device__ __forceinline__ RayLds ray_lds(float* base, int r, int D) {
    float* b = base + r * RAY_LDS_FLOATS(D);
    RayLds L;
    L.dc = b; L.sc = b + D; L.df = b + 2 * D; L.sf = b + 3 * D;
    L.sd = b + 4 * D; L.ss = b + 6 * D; L.w = b + 8 * D; L.q = b + 10 * D; L.t = b + 12 * D; L.misc = b + 14 * D;
    return L;
}

// float min / max as native integer atomics on the value's bit pattern (atomicMin / atomicMax on float compile to compare-and-swap
// loops, and 4096 workgroups looping on the same two addresses cost ~0.3 ms per launch): non-negative floats order like signed ints,
// negative ones like reversed unsigned ints.  The cell holds an ordinary float throughout (initialised to +inf / -inf by the caller).
__device__ __forceinline__ void atomic_min_float(float* a, float v) {
    v += 0.0f;                                        // -0 -> +0
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(a), __float_as_int(v));
    else if (v < 0.f) atomicMax(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_float(float* a, float v) {
    v += 0.0f;
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(a), __float_as_int(v));
    else if (v < 0.f) atomicMin(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}

// weights of nS sorted samples (depths d, densities sg): alpha_i -> q_i -> T_i -> w_i, serial scan by one thread per ray
#ifndef RK_FAST_ALPHA
#define RK_FAST_ALPHA 1
#endif
__device__ __forceinline__ void march_alpha(const float* d, const float* sg, int i, float& alpha, float& delta, float& dens_mid) {
    delta = d[i + 1] - d[i];
    dens_mid = (sg[i] + sg[i + 1]) * 0.5f;
#if RK_FAST_ALPHA
    // hardware exp2 / log2 forms (absolute error < 2e-7 on sp and on alpha -- the size of the rounding of `1 - exp(-x)` itself); the libm
    // expf / log1pf pair is ~450 of the ~750 vector instructions a thread of render_kernel<3> spends per interval
    const float sp = softplusf_(dens_mid - 1.f);
    alpha = 1.f - __expf(-(sp * delta));
#else
    float sp = softplus_acc(dens_mid - 1.f);
    alpha = 1.f - expf(-(sp * delta));
#endif
}

// ---- the per-ray serial scans ---------------------------------------------------------------------------------------------------------
// Transmittance scan by ONE lane per ray: w_i = alpha_i T_i, T_{i+1} = T_i q_i, (+ sum w, sum w mid_i), in interval order -- the order of the
// reference's cumprod / sum on its CPU path, kept so that results do not change.  The recurrence is one multiply per interval; what made the
// loop slow was its shape: four LDS reads, their full latency, and a store PER INTERVAL (the arrays may alias, so the compiler neither batches
// nor hoists them) while the block's other waves wait at the barrier -- SQ_WAIT_ANY was 63 % of the wave cycles of render_kernel<1>.  With
// `vec` (16-byte aligned per-ray arrays: D even) the operands of eight intervals arrive as 16-byte reads ahead of the chain and leave as
// 16-byte stores: ~12 LDS round trips per scan instead of ~95.  Arrays are 2 D long; a tail batch reads (and zero-fills) up to index 2 D - 1.
__device__ __forceinline__ float4 lds_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void lds_st4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }

template <bool KEEP_T, bool SUMS, int B>          // B = intervals per batch (4 or 8: the batch lives in registers of EVERY wave of the kernel)
__device__ __forceinline__ void march_scan(const RayLds& Ls, int nI, int cap, bool vec, float& wsum_o, float& dnum_o) {
    float T = 1.f, wsum = 0.f, dnum = 0.f;
    if (vec) {
        for (int i0 = 0; i0 < nI; i0 += B) {
            const bool full = i0 + B <= cap;                     // (cap = array length: a last batch that would run past it goes one by one)
            float tv[B], qv[B], sv[B + 1], Tv[B];
            if (full) {
#pragma unroll
                for (int g = 0; g < B / 4; ++g) {
                    const float4 ta = lds_ld4(Ls.t + i0 + 4 * g), qa = lds_ld4(Ls.q + i0 + 4 * g);
                    tv[4 * g] = ta.x; tv[4 * g + 1] = ta.y; tv[4 * g + 2] = ta.z; tv[4 * g + 3] = ta.w;
                    qv[4 * g] = qa.x; qv[4 * g + 1] = qa.y; qv[4 * g + 2] = qa.z; qv[4 * g + 3] = qa.w;
                    if (SUMS) {
                        const float4 sa = lds_ld4(Ls.sd + i0 + 4 * g);
                        sv[4 * g] = sa.x; sv[4 * g + 1] = sa.y; sv[4 * g + 2] = sa.z; sv[4 * g + 3] = sa.w;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const bool in = i0 + k < cap;
                    tv[k] = in ? Ls.t[i0 + k] : 0.f; qv[k] = in ? Ls.q[i0 + k] : 1.f;
                    if (SUMS) sv[k] = in ? Ls.sd[i0 + k] : 0.f;
                }
            }
            if (SUMS) sv[B] = i0 + B < cap ? Ls.sd[i0 + B] : 0.f;
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const bool ok = i0 + k < nI;
                const float w = tv[k] * T;
                tv[k] = ok ? w : 0.f; Tv[k] = T;
                if (ok) {
                    T *= qv[k];
                    if (SUMS) { wsum += w; dnum += w * (0.5f * (sv[k] + sv[k + 1])); }
                }
            }
            if (full) {
#pragma unroll
                for (int g = 0; g < B / 4; ++g) {
                    lds_st4(Ls.w + i0 + 4 * g, tv[4 * g], tv[4 * g + 1], tv[4 * g + 2], tv[4 * g + 3]);
                    if (KEEP_T) lds_st4(Ls.t + i0 + 4 * g, Tv[4 * g], Tv[4 * g + 1], Tv[4 * g + 2], Tv[4 * g + 3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < B; ++k)
                    if (i0 + k < nI) { Ls.w[i0 + k] = tv[k]; if (KEEP_T) Ls.t[i0 + k] = Tv[k]; }
            }
        }
    } else {
        for (int i = 0; i < nI; ++i) {
            const float w = Ls.t[i] * T;
            Ls.w[i] = w;
            if (KEEP_T) Ls.t[i] = T;                  // keep T_i for the gradient
            T *= Ls.q[i];
            if (SUMS) { wsum += w; dnum += w * (0.5f * (Ls.sd[i] + Ls.sd[i + 1])); }
        }
    }
    wsum_o = wsum; dnum_o = dnum;
}

// Exclusive suffix sums of e[0 .. n) in place, from the far end (the order of the backward's reverse scan): e[i] <- sum_{j > i} e[j]
__device__ __forceinline__ void suffix_scan_inplace(float* e, int n, int cap, bool vec) {
    float S = 0.f;
    if (vec) {
        for (int i0 = ((n - 1) >> 3) << 3; i0 >= 0; i0 -= 8) {
            const bool full = i0 + 8 <= cap;
            float ev[8];
            if (full) {
                const float4 a = lds_ld4(e + i0), b = lds_ld4(e + i0 + 4);
                ev[0] = a.x; ev[1] = a.y; ev[2] = a.z; ev[3] = a.w; ev[4] = b.x; ev[5] = b.y; ev[6] = b.z; ev[7] = b.w;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) ev[k] = i0 + k < n ? e[i0 + k] : 0.f;
            }
#pragma unroll
            for (int k = 7; k >= 0; --k) {
                if (i0 + k < n) { const float t = ev[k]; ev[k] = S; S += t; }
            }
            if (full) { lds_st4(e + i0, ev[0], ev[1], ev[2], ev[3]); lds_st4(e + i0 + 4, ev[4], ev[5], ev[6], ev[7]); }
            else {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (i0 + k < n) e[i0 + k] = ev[k];
            }
        }
    } else {
        for (int i = n - 1; i >= 0; --i) { const float t = e[i]; e[i] = S; S += t; }
    }
}

// rank of `x` among `n` LDS values: the number that sort before it -- v < x, or (ties == 1) v <= x, or (ties == 2) v == x with index < self
__device__ __forceinline__ int count_before(const float* v, int n, float x, int ties, int self, bool vec) {
    int cnt = 0, i = 0;
    if (vec) {
        for (; i + 4 <= n; i += 4) {
            const float4 o = lds_ld4(v + i);
            const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) cnt += (ov[k] < x || (ov[k] == x && (ties == 1 || (ties == 2 && i + k < self)))) ? 1 : 0;
        }
    }
    for (; i < n; ++i) { const float o = v[i]; cnt += (o < x || (o == x && (ties == 1 || (ties == 2 && i < self)))) ? 1 : 0; }
    return cnt;
}

// The colour reduction of the compositing stage goes through LDS CCH channels at a time: a full [threads][33] buffer (25 KB) holds a
// block to 3 waves per SIMD; [threads][9] leaves room for 6 (the stage streams 200 MB of saved rows and is latency-bound).
#ifndef RK_CCH
#define RK_CCH 16         // 8: 75.8 us, 16: 72.2 us, 32: 84.8 us for render_kernel<3> (LDS per block vs barriers per ray)
#endif
constexpr int CCH = RK_CCH;
__host__ __device__ constexpr int render_red_floats(int RPB, int D, int mode) {
    return mode == 1 ? 2 * RPB * 2 * D + RPB * D * 7 : (mode == 2 ? 0 : RPB * D * (CCH + 1));
}

// MODE 0: fused forward (decoder on the vector ALUs inside the ray kernel).  MODE 1: ray-level backward.  MODE 2 / 3: the two ray-level
// stages of the pipelined forward -- 2 = importance sampling from the saved coarse densities (writes fine depths and fine sample
// positions), 3 = merge + march + compositing from the saved (sigma, colour) rows of both passes.
template <int MODE>
__global__ void __launch_bounds__(MAXT) render_kernel(const eg3d_render_bwd_params bp) {
    constexpr bool BWD = MODE == 1;
    const eg3d_render_params& p = bp.fwd;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int D = p.Dc > p.Df ? p.Dc : p.Df;
    const int RPB = MAXT / D;                    // rays per block
#ifndef RK_VEC_SCAN
#define RK_VEC_SCAN 1          // batched transmittance scan: 1 = backward kernel only (-1.7 us; the forward kernels measured no gain, render_kernel<3> lost occupancy), 2 = everywhere
#endif
#ifndef RK_SORTED_FAST
#define RK_SORTED_FAST 1
#endif
#ifndef RK_VEC_RANK
#define RK_VEC_RANK 0          // 16-byte reads in the merge's rank loops: measured +5 .. +10 us on render_kernel<3> (104 registers: four waves per SIMD instead of five), off
#endif
    const bool vec = (D & 1) == 0;               // per-ray LDS arrays start on 16-byte boundaries (RAY_LDS_FLOATS(D) * 4 and the 2 D / 4 D offsets are multiples of 16)
    const int nthreads = RPB * D;
    const int tid = threadIdx.x;
    const int r = tid / D, s = tid - r * D;
    const int64_t nrays = (int64_t)p.N * p.R;
    const int nblocks = (int)((nrays + RPB - 1) / RPB);
    const int bid = eg3d_xcd_remap(blockIdx.x, nblocks);
    const int64_t slot = (int64_t)bid * RPB + r;
    const bool live = tid < nthreads && slot < nrays;
    const int64_t ray = live ? eg3d_render::ray_of_slot(slot, p.R, p.ray_tile_width) : slot;
    const int64_t rr = live ? ray : 0;
    const int n = (int)(rr / p.R);
    RayLds L = ray_lds(lds, r < RPB ? r : 0, D);
    // the serial scans of the block's RPB rays run on lanes 0 .. RPB-1 of wave 0 (one lane per ray at thread s == 0 put them into three
    // different waves, each of which then issued the whole loop for one or two active lanes)
    const bool scan_lane = tid < RPB && (int64_t)bid * RPB + tid < nrays;
    const RayLds Ls = ray_lds(lds, scan_lane ? tid : 0, D);
    float* red = lds + RPB * RAY_LDS_FLOATS(D);   // forward: [nthreads][33] colour reduction; backward: E / GA / coordinate sums
    float* grgb = red + render_red_floats(RPB, D, MODE) + (r < RPB ? r : 0) * CO;   // backward: this ray's incoming d_rgb [32]

    const float ox = p.origins[rr * 3 + 0], oy = p.origins[rr * 3 + 1], oz = p.origins[rr * 3 + 2];
    const float dx = p.dirs[rr * 3 + 0], dy = p.dirs[rr * 3 + 1], dz = p.dirs[rr * 3 + 2];
    const float cs = 2.f / p.box_warp;
    const float* pn = p.planes + (int64_t)n * p.Hp * p.Wp * p.ldp;
    const int Dc = p.Dc, Df = p.Df;
    const bool has_c = live && s < Dc, has_f = live && s < Df;

    if (BWD) {
        if (live) for (int k = s; k < CO; k += D) grgb[k] = bp.d_rgb[rr * CO + k];
        __syncthreads();
    }

    // ---------------- coarse sample ----------------
    // forward keeps the 32 colours of both samples in registers until the weights are known; backward only needs
    // e = <d_rgb, colour> per sample at this point (the MLP is re-run in the per-sample gradient pass).
    constexpr int NKEEP = (BWD || MODE == 2) ? 1 : CO;
    float depth_c = 0.f, sig_c = 0.f, rgb_c[NKEEP], e_c = 0.f;
#pragma unroll
    for (int k = 0; k < NKEEP; ++k) rgb_c[k] = 0.f;
    if constexpr (BWD) if (has_c) {      // ray-level backward: (sigma, colour) come from the forward's save buffers
        depth_c = coarse_depth(p, rr, s, p.u1[rr * Dc + s]);
        const int64_t row = (rr * 2 + 0) * D + s;
        sig_c = p.save_sigma[row];
        const float4* c4p = reinterpret_cast<const float4*>(p.save_rgb + row * CO);
#pragma unroll
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float4 v = c4p[c4];
            e_c = fmaf(grgb[c4 * 4], v.x, e_c); e_c = fmaf(grgb[c4 * 4 + 1], v.y, e_c);
            e_c = fmaf(grgb[c4 * 4 + 2], v.z, e_c); e_c = fmaf(grgb[c4 * 4 + 3], v.w, e_c);
        }
        L.dc[s] = depth_c; L.sc[s] = sig_c;
    }
    if constexpr (MODE >= 2) if (has_c) {      // pipelined forward: the sample-level kernel already decoded the coarse rows
        depth_c = coarse_depth(p, rr, s, p.u1[rr * Dc + s]);
        const int64_t row = (rr * 2 + 0) * D + s;
        sig_c = p.save_sigma[row];
        if constexpr (MODE == 3) {          // (reading the colour rows only where they are composited frees 64 registers across the merge and the scan -- and
            // exposes their latency: 76 -> 118 us)
            const float4* c4p = reinterpret_cast<const float4*>(p.save_rgb + row * CO);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) {
                float4 v = c4p[c4];
                rgb_c[c4 * 4] = v.x; rgb_c[c4 * 4 + 1] = v.y; rgb_c[c4 * 4 + 2] = v.z; rgb_c[c4 * 4 + 3] = v.w;
            }
        }
        L.dc[s] = depth_c; L.sc[s] = sig_c;
    }
    if constexpr (MODE == 0) if (has_c) {
        float f[FC], out[1 + CO];
        depth_c = coarse_depth(p, rr, s, p.u1[rr * Dc + s]);
        gather_feats(pn, p.Hp, p.Wp, p.ldp, cs, ox + depth_c * dx, oy + depth_c * dy, oz + depth_c * dz, f);   // mul+add like the reference (no fma)
        mlp_fwd(p.w0, p.b0, p.w1, p.b1, f, out);
        sig_c = out[0];
#pragma unroll
        for (int k = 0; k < CO; ++k) rgb_c[k] = sigmoidf_(out[1 + k]) * 1.002f - 0.001f;
        if (p.save_sigma) {                 // training mode: keep (sigma, colour) per sample for the backward
            const int64_t row = (rr * 2 + 0) * D + s;
            p.save_sigma[row] = sig_c;
            float4* o = reinterpret_cast<float4*>(p.save_rgb + row * CO);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) o[c4] = make_float4(rgb_c[c4 * 4], rgb_c[c4 * 4 + 1], rgb_c[c4 * 4 + 2], rgb_c[c4 * 4 + 3]);
        }
        L.dc[s] = depth_c; L.sc[s] = sig_c;
    }
    __syncthreads();

    // ---------------- importance sampling (forward only; backward re-reads the saved fine depths) ----------------
    float depth_f = 0.f;
    if (Df > 0) {
        if (MODE == 0 || MODE == 2) {
            // coarse march -> weights (Dc-1 intervals)
            if (live && s < Dc - 1) {
                float a, de, dm;
                march_alpha(L.dc, L.sc, s, a, de, dm);
                L.t[s] = a; L.q[s] = 1.f - a + 1e-10f;
            }
            __syncthreads();
            if (scan_lane) {
                float unused0, unused1;
                march_scan<false, false, (MODE == 0 ? 4 : 8)>(Ls, Dc - 1, 2 * D, vec && RK_VEC_SCAN, unused0, unused1);
            }
            __syncthreads();
            // smoothing: max_pool1d(2,1,pad 1) -> avg_pool1d(2,1) -> + 0.01      (renderer.py:260-262)
            const int nw = Dc - 1;
            float sm = 0.f;
            if (live && s < nw) {
                float wl = s > 0 ? L.w[s - 1] : -INFINITY, wc = L.w[s], wr = s + 1 < nw ? L.w[s + 1] : -INFINITY;
                sm = 0.5f * (fmaxf(wl, wc) + fmaxf(wc, wr)) + 0.01f;
            }
            __syncthreads();
            if (live && s < nw) L.t[s] = sm;
            __syncthreads();
            // sample_pdf on bins = mids[0..nw-1], weights = sm[1..nw-2]   (renderer.py:264-266,281-307)
            const int ns = nw - 2;
            const float eps = 1e-5f;
            float tot = 0.f;
            if (live) for (int k = 0; k < ns; ++k) tot += L.t[1 + k] + eps;
            // pdf_k = (sm_k + eps) / tot once per ray (thread k), not inside every thread's search: the division is ten dependent instructions
            // per step of a loop that the 48 threads of a ray all walk (q[] is free: the coarse scan has consumed it)
            if (live && s >= 1 && s <= ns) L.q[s] = (L.t[s] + eps) / tot;
            __syncthreads();
            if (has_f) {
                const float u = p.u2[rr * Df + s];
                float c = 0.f, c_below = 0.f, c_above = 0.f;
                int inds = ns + 1;
                for (int k = 1; k <= ns; ++k) {
                    float cn = c + L.q[k];
                    if (cn > u) { inds = k; c_below = c; c_above = cn; c = cn; break; }
                    c = cn;
                }
                int below, above;
                if (inds > ns) { below = ns; above = ns; c_below = c; c_above = c; }
                else { below = inds - 1; above = inds; }
                float bb = 0.5f * (L.dc[below] + L.dc[below + 1]), ba = 0.5f * (L.dc[above] + L.dc[above + 1]);
                float denom = c_above - c_below;
                if (denom < eps) denom = 1.f;
                depth_f = bb + (u - c_below) / denom * (ba - bb);
                if (p.fine_depths) p.fine_depths[rr * Df + s] = depth_f;
                if (p.dbg_inds) { int32_t* q = p.dbg_inds + (rr * Df + s) * 3; q[0] = inds; q[1] = below; q[2] = above; }
                if (p.dbg_cdf && s == 0) {
                    float cc = 0.f;
                    for (int k = 1; k <= ns; ++k) { cc = cc + (L.t[k] + eps) / tot; p.dbg_cdf[rr * ns + k - 1] = cc; }
                }
            }
        } else {
            if (has_f) depth_f = p.fine_depths[rr * Df + s];
        }
    }
    if constexpr (MODE == 2) {          // hand the fine sample positions to the sample-level kernel and stop here
        if (live) {
            float4 ps = make_float4(NAN, 0.f, 0.f, 0.f);
            if (has_f) ps = make_float4(ox + depth_f * dx, oy + depth_f * dy, oz + depth_f * dz, depth_f);
            reinterpret_cast<float4*>(p.pos_rows)[((int64_t)nrays + rr) * D + s] = ps;
        }
        return;
    }

    // ---------------- fine sample ----------------
    float sig_f = 0.f, rgb_f[NKEEP], e_f = 0.f;
#pragma unroll
    for (int k = 0; k < NKEEP; ++k) rgb_f[k] = 0.f;
    if constexpr (BWD) if (has_f) {
        const int64_t row = (rr * 2 + 1) * D + s;
        sig_f = p.save_sigma[row];
        const float4* c4p = reinterpret_cast<const float4*>(p.save_rgb + row * CO);
#pragma unroll
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float4 v = c4p[c4];
            e_f = fmaf(grgb[c4 * 4], v.x, e_f); e_f = fmaf(grgb[c4 * 4 + 1], v.y, e_f);
            e_f = fmaf(grgb[c4 * 4 + 2], v.z, e_f); e_f = fmaf(grgb[c4 * 4 + 3], v.w, e_f);
        }
        L.df[s] = depth_f; L.sf[s] = sig_f;
    }
    if constexpr (MODE == 3) if (has_f) {
        const int64_t row = (rr * 2 + 1) * D + s;
        sig_f = p.save_sigma[row];
        const float4* c4p = reinterpret_cast<const float4*>(p.save_rgb + row * CO);
#pragma unroll
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float4 v = c4p[c4];
            rgb_f[c4 * 4] = v.x; rgb_f[c4 * 4 + 1] = v.y; rgb_f[c4 * 4 + 2] = v.z; rgb_f[c4 * 4 + 3] = v.w;
        }
        L.df[s] = depth_f; L.sf[s] = sig_f;
    }
    if constexpr (MODE == 0) if (has_f) {
        float f[FC], out[1 + CO];
        gather_feats(pn, p.Hp, p.Wp, p.ldp, cs, ox + depth_f * dx, oy + depth_f * dy, oz + depth_f * dz, f);
        mlp_fwd(p.w0, p.b0, p.w1, p.b1, f, out);
        sig_f = out[0];
#pragma unroll
        for (int k = 0; k < CO; ++k) rgb_f[k] = sigmoidf_(out[1 + k]) * 1.002f - 0.001f;
        if (p.save_sigma) {
            const int64_t row = (rr * 2 + 1) * D + s;
            p.save_sigma[row] = sig_f;
            float4* o = reinterpret_cast<float4*>(p.save_rgb + row * CO);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) o[c4] = make_float4(rgb_f[c4 * 4], rgb_f[c4 * 4 + 1], rgb_f[c4 * 4 + 2], rgb_f[c4 * 4 + 3]);
        }
        L.df[s] = depth_f; L.sf[s] = sig_f;
    }
    __syncthreads();

    // ---------------- merge (stable: coarse precedes fine at ties) ----------------
    const int nS = Dc + Df;
    int rank_c = s, rank_f = 0;
    // position in torch.sort(cat(coarse, fine), stable=True) (renderer.py:212-222).  The stratified depths t_i + u_i * delta are increasing
    // in exact arithmetic only: u = 1 - 2^-24 next to u = 0 can round the wrong way round, so the coarse list is ranked too, not assumed sorted --
    // but it is CHECKED sorted first (one compare per thread, one block-wide OR): when every ray of the block has strictly increasing coarse
    // depths (all but a handful of blocks per million rays) a coarse sample's rank among the coarse ones is its index and a fine sample's is an
    // upper bound found by bisection: 102 instead of 192 LDS reads per thread, the same integers (test_sampler_indices_exact)
#if RK_SORTED_FAST
    const int unsorted = __syncthreads_or((has_c && s + 1 < Dc && !(depth_c < L.dc[s + 1])) ? 1 : 0);
#else
    const int unsorted = 1;
#endif
    if (has_c) {
        rank_c = (unsorted ? count_before(L.dc, Dc, depth_c, 2, s, vec && RK_VEC_RANK) : s) + count_before(L.df, Df, depth_c, 0, 0, vec && RK_VEC_RANK);
        L.sd[rank_c] = depth_c; L.ss[rank_c] = sig_c;
    }
    if (has_f) {
        int below;
        if (unsorted) below = count_before(L.dc, Dc, depth_f, 1, 0, vec && RK_VEC_RANK);
        else {                                      // number of coarse depths <= depth_f (NaN: 0, as the count)
            int lo = 0, hi = Dc;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.dc[mid] <= depth_f) lo = mid + 1; else hi = mid; }
            below = lo;
        }
        rank_f = below + count_before(L.df, Df, depth_f, 2, s, vec && RK_VEC_RANK);
        L.sd[rank_f] = depth_f; L.ss[rank_f] = sig_f;
    }
    if (p.dbg_ranks) {
        if (has_c) p.dbg_ranks[rr * nS + s] = rank_c;
        if (has_f) p.dbg_ranks[rr * nS + Dc + s] = rank_f;
    }
    __syncthreads();

    // ---------------- final march over nS-1 intervals ----------------
    const int nI = nS - 1;
    if (live) {
        for (int i = s; i < nI; i += D) {
            float a, de, dm;
            march_alpha(L.sd, L.ss, i, a, de, dm);
            L.t[i] = a; L.q[i] = 1.f - a + 1e-10f;
        }
    }
    __syncthreads();
    if (scan_lane) {
        float wsum, dnum;
        march_scan<BWD, true, (BWD ? 8 : 4)>(Ls, nI, 2 * D, vec && RK_VEC_SCAN && (BWD || RK_VEC_SCAN > 1), wsum, dnum);          // (backward: T_i kept in t[] for the gradient)
        Ls.misc[0] = wsum; Ls.misc[1] = dnum; Ls.misc[2] = Ls.sd[0]; Ls.misc[3] = Ls.sd[nS - 1];
    }
    __syncthreads();

    if constexpr (MODE == 0 || MODE == 3) {
        // composite colour: each thread contributes a_c * rgb_c + a_f * rgb_f, reduced over the ray's D threads via LDS
        float a_c = 0.f, a_f = 0.f;
        if (has_c) a_c = 0.5f * ((rank_c > 0 ? L.w[rank_c - 1] : 0.f) + (rank_c < nI ? L.w[rank_c] : 0.f));
        if (has_f) a_f = 0.5f * ((rank_f > 0 ? L.w[rank_f - 1] : 0.f) + (rank_f < nI ? L.w[rank_f] : 0.f));
#pragma unroll
        for (int c0 = 0; c0 < NKEEP; c0 += CCH) {
            if (tid < nthreads) {
                float* my = red + tid * (CCH + 1);
#pragma unroll
                for (int k = 0; k < CCH; ++k) my[k] = a_c * rgb_c[c0 + k] + a_f * rgb_f[c0 + k];
            }
            __syncthreads();
            if (live) {
                for (int k = s; k < CCH; k += D) {
                    float acc = 0.f;
                    const float* col = red + (r * D) * (CCH + 1) + k;
                    for (int j = 0; j < D; ++j) acc += col[j * (CCH + 1)];
                    float wsum = L.misc[0];
                    if (p.white_back) acc = acc + 1.f - wsum;
                    p.rgb[rr * CO + c0 + k] = acc * 2.f - 1.f;
                }
            }
            __syncthreads();
        }
        if (live && s == 0) {
            float wsum = L.misc[0];
            p.depth[rr] = L.misc[1] / wsum;          // NaN when wsum == 0; finalize handles it
            p.wsum[rr] = wsum;
        }
        // global depth range (ray_marcher.py:50 clamps to min/max over ALL samples of ALL rays)
        __syncthreads();
        if (tid == 0) {
            float mn = INFINITY, mx = -INFINITY;
            for (int q = 0; q < RPB; ++q) {
                if ((int64_t)bid * RPB + q < nrays) {
                    RayLds Lq = ray_lds(lds, q, D);
                    mn = fminf(mn, Lq.misc[2]); mx = fmaxf(mx, Lq.misc[3]);
                }
            }
            // look first: the range is monotone, so a stale read only costs a redundant atomic -- and after the first few workgroups
            // almost nobody improves it (4096 workgroups x 2 same-address atomics, ~30 ns apart, is what the atomic unit sustains at best)
            if (mn < __hip_atomic_load(p.depth_minmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_min_float(p.depth_minmax, mn);
            if (mx > __hip_atomic_load(p.depth_minmax + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_float(p.depth_minmax + 1, mx);
        }
        return;
    }

    // =====================================================================================================
    // backward
    // =====================================================================================================
    float* E = red + r * 2 * D;                  // [nS] <d_rgb, colour> at the sorted position of each sample
    float* GA = red + RPB * 2 * D + r * 2 * D;   // [nI] dL/d(mid density) per interval
    float* GWT = red + 2 * RPB * 2 * D + r * 4 * D;      // [nI] gw_i T_i     (scratch behind E / GA: render_red_floats keeps 7 D floats per ray there)
    float* DL = GWT + 2 * D;                     // [nI] delta_i
    if (has_c) E[rank_c] = e_c;
    if (has_f) E[rank_f] = e_f;
    __syncthreads();
    // Everything of an interval that does not depend on the suffix sum is formed by the ray's D threads, two intervals each.  The serial scan
    // below used to do all of it -- sigmoid, divisions, five LDS reads per interval -- on ONE lane per ray: 112 M wave-level VALU instructions
    // per launch (9 K per wave), i.e. the kernel was bound by arithmetic that 63 of 64 lanes sat out.
    float xw[2] = {0.f, 0.f};
    if (live) {
        const float wsum = L.misc[0], dnum = L.misc[1];
        const float depth_raw = dnum / wsum;
        const float dmin = p.depth_minmax[0], dmax = p.depth_minmax[1];
        float gd = bp.d_depth ? bp.d_depth[rr] : 0.f;
        if (!(depth_raw >= dmin && depth_raw <= dmax)) gd = 0.f;        // NaN / clamped: no gradient
        const float gws = bp.d_wsum ? bp.d_wsum[rr] : 0.f;
        float grgb_sum = 0.f;
        if (p.white_back) for (int k = 0; k < CO; ++k) grgb_sum += grgb[k];
        int k = 0;
        for (int i = s; i < nI; i += D, ++k) {
            const float dmid = 0.5f * (L.sd[i] + L.sd[i + 1]);
            float gw = (E[i] + E[i + 1]) + gws - 2.f * grgb_sum;
            if (gd != 0.f) gw += gd * (dmid - depth_raw) / wsum;
            // alpha = 1 - exp(-sp*delta);  sp = softplus(m - 1)
            const float m = 0.5f * (L.ss[i] + L.ss[i + 1]) - 1.f;
            GWT[i] = gw * L.t[i];
            DL[i] = L.sd[i + 1] - L.sd[i];
            GA[i] = m > 20.f ? 1.f : (RK_FAST_ALPHA ? sigmoidf_(m) : sigmoid_acc(m));         // replaced by the interval's gradient in the scan
            xw[k] = gw * L.w[i];
        }
    }
    __syncthreads();                 // every E[i], E[i + 1] has been read: E becomes the scan's addend gw_i w_i
    if (live) {
        int k = 0;
        for (int i = s; i < nI; i += D, ++k) E[i] = xw[k];
    }
    __syncthreads();
    // dL/dalpha_i = gw_i T_i - S_i / q_i with the reverse suffix sum S_i = sum_{j > i} gw_j w_j: only the sum itself is a recurrence (one add per
    // interval, one lane per ray); the division and the products around it are formed by the ray's D threads again
    if (scan_lane) suffix_scan_inplace(red + tid * 2 * D, nI, 2 * D, vec);
    __syncthreads();
    if (live) {
        for (int i = s; i < nI; i += D) {
            const float q = L.q[i];
            const float galpha = GWT[i] - E[i] / q;
            const float one_minus_alpha = q - 1e-10f;
            const float gsp = galpha * DL[i] * one_minus_alpha;
            GA[i] = gsp * GA[i];
        }
    }
    __syncthreads();

    if (bp.df_amax != nullptr && blockIdx.x == 0 && tid == 0) *bp.df_amax = 0.f;      // the sample-level kernel (next launch) accumulates max|df_rows| here
    // position rows (x, y, z, depth) of this thread's two samples: input of the sample-level kernel and of the plane scatter
    if (live && s < D) {
        float4* pr = reinterpret_cast<float4*>(bp.df_pos);
        pr[(rr * 2 + 0) * D + s] = has_c ? make_float4(ox + depth_c * dx, oy + depth_c * dy, oz + depth_c * dz, depth_c) : make_float4(NAN, 0, 0, 0);
        pr[(rr * 2 + 1) * D + s] = has_f ? make_float4(ox + depth_f * dx, oy + depth_f * dy, oz + depth_f * dz, depth_f) : make_float4(NAN, 0, 0, 0);
    }
    // per-sample results of the ray-level pass: a = colour weight (dL/d colour = 2 a d_rgb), gsig = dL/d sigma
    if (has_c) {
        const float a = 0.5f * ((rank_c > 0 ? L.w[rank_c - 1] : 0.f) + (rank_c < nI ? L.w[rank_c] : 0.f));
        const float gs = 0.5f * ((rank_c > 0 ? GA[rank_c - 1] : 0.f) + (rank_c < nI ? GA[rank_c] : 0.f));
        reinterpret_cast<float2*>(bp.ag_rows)[(rr * 2 + 0) * D + s] = make_float2(a, gs);
    }
    if (has_f) {
        const float a = 0.5f * ((rank_f > 0 ? L.w[rank_f - 1] : 0.f) + (rank_f < nI ? L.w[rank_f] : 0.f));
        const float gs = 0.5f * ((rank_f > 0 ? GA[rank_f - 1] : 0.f) + (rank_f < nI ? GA[rank_f] : 0.f));
        reinterpret_cast<float2*>(bp.ag_rows)[(rr * 2 + 1) * D + s] = make_float2(a, gs);
    }
}

// d_origins[ray] = sum_rows g;  d_dirs[ray] = sum_rows depth * g
__global__ void __launch_bounds__(256) render_coord_reduce_kernel(const float4* __restrict__ gc, float* __restrict__ d_o, float* __restrict__ d_d,
                                                                  int64_t nrays, int rows_per_ray) {
    const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= nrays) return;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    const float4* g = gc + ray * rows_per_ray;
    for (int i = 0; i < rows_per_ray; ++i) {
        const float4 v = g[i];
        o0 += v.x; o1 += v.y; o2 += v.z;
        d0 += v.x * v.w; d1 += v.y * v.w; d2 += v.z * v.w;
    }
    if (d_o) { d_o[ray * 3] = o0; d_o[ray * 3 + 1] = o1; d_o[ray * 3 + 2] = o2; }
    if (d_d) { d_d[ray * 3] = d0; d_d[ray * 3 + 1] = d1; d_d[ray * 3 + 2] = d2; }
}

// =========================================================================================================
// Tile-binned scatter of the per-sample feature gradients into the tri-plane gradient.
//   A float-atomic scatter costs 12 corners x 32 channels = 384 atomics per sample (604 M per backward at the FFHQ
//   config; measured 30 ms on MI355X = the atomic rate of the memory fabric).  Instead: (1) count the (sample, plane)
//   pairs per 16x16-texel tile, (2) exclusive scan, (3) counting-sort the pair ids by tile, (4) one block per tile
//   accumulates its pairs into a 17x17x32 LDS tile with conflict-free ds_add_f32 (lane = channel) and flushes the tile
//   once.  Global atomics drop to ~9 K per tile.
// =========================================================================================================
constexpr int TS = 15;          // interior texels per tile side; a tile's accumulator covers TS + 1 = 16 rows / columns (the last one is shared with the neighbour)

__device__ __forceinline__ bool plane_cell(const float4 pos, int pl, float cs, int Hp, int Wp, int& x0, int& y0, float& wx1, float& wy1) {
    float u, v;
    plane_uv(pl, pos.x * cs, pos.y * cs, pos.z * cs, u, v);
    float ix = ((u + 1.f) * Wp - 1.f) * 0.5f, iy = ((v + 1.f) * Hp - 1.f) * 0.5f;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    wx1 = ix - fx0; wy1 = iy - fy0;
    if (!(fx0 >= -1.f && fx0 <= (float)(Wp - 1) && fy0 >= -1.f && fy0 <= (float)(Hp - 1))) return false;   // no corner in range (also NaN)
    x0 = (int)fx0; y0 = (int)fy0;
    return true;
}

__device__ __forceinline__ int tile_of(int x0, int y0, int ntx, int nty) {
    int tx = (x0 < 0 ? 0 : x0) / TS, ty = (y0 < 0 ? 0 : y0) / TS;
    return (ty < nty ? ty : nty - 1) * ntx + (tx < ntx ? tx : ntx - 1);
}

// pass 0: count; pass 1: place.  One thread per (sample row, plane).  Bin counters are pre-aggregated in an LDS histogram so
// that the global counters see one atomic per (block, touched bin) instead of one per pair (the pairs of neighbouring rows
// fall into the same few tiles: same-address atomics would serialise).
constexpr int BIN_ITEMS = 16;           // pairs per thread (block-strided).  4 per thread took 87 + 99 us for the two passes, 16: 36 + 41 -- the per-block flush
                                        // (an LDS sweep + global atomics for every touched bin) is the cost, not the per-pair work; 32 / 64 are slower again
template <int PASS>
__global__ void __launch_bounds__(256) scatter_bin_kernel(const float4* __restrict__ pos, int64_t S, int64_t rows_per_image, float cs, int Hp, int Wp,
                                                          int ntx, int nty, int nb, int* __restrict__ counts, const int* __restrict__ offsets,
                                                          int* __restrict__ fill, int* __restrict__ ids) {
    extern __shared__ int hist[];       // [nb] local counts, then (pass 1) [nb] reserved bases
    for (int i = threadIdx.x; i < nb; i += 256) hist[i] = 0;
    __syncthreads();
    int bin[BIN_ITEMS], lrank[BIN_ITEMS], rowi[BIN_ITEMS];
    const int64_t base = (int64_t)blockIdx.x * 256 * BIN_ITEMS + threadIdx.x;          // consecutive lanes -> consecutive (row, plane) pairs
#pragma unroll
    for (int k = 0; k < BIN_ITEMS; ++k) {
        bin[k] = -1;
        const int64_t i = base + (int64_t)k * 256;
        if (i >= S * 3) continue;
        const int64_t row = i / 3;
        const int pl = (int)(i - row * 3);
        const float4 ps = pos[row];
        if (isnan(ps.x)) continue;
        int x0, y0; float wx1, wy1;
        if (!plane_cell(ps, pl, cs, Hp, Wp, x0, y0, wx1, wy1)) continue;
        const int n = (int)(row / rows_per_image);
        bin[k] = (n * 3 + pl) * (ntx * nty) + tile_of(x0, y0, ntx, nty);
        rowi[k] = (int)row;
        lrank[k] = atomicAdd(&hist[bin[k]], 1);
    }
    __syncthreads();
    if constexpr (PASS == 0) {
        for (int i = threadIdx.x; i < nb; i += 256) { int c = hist[i]; if (c) atomicAdd(counts + i, c); }
    } else {
        int* basep = hist + nb;
        for (int i = threadIdx.x; i < nb; i += 256) { int c = hist[i]; basep[i] = c ? offsets[i] + atomicAdd(fill + i, c) : 0; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BIN_ITEMS; ++k)
            if (bin[k] >= 0) {
                if (PASS == 1) {
                    ids[basep[bin[k]] + lrank[k]] = rowi[k] << 2;       // row id (the plane is implied by the bin)
                } else {            // PASS 2: the whole pair record, so that the accumulate pass streams it (no id -> position gather there)
                    const int ntile = ntx * nty, t = bin[k] % ntile, pl = (bin[k] / ntile) % 3;
                    int x0, y0; float wx1, wy1;
                    plane_cell(pos[rowi[k]], pl, cs, Hp, Wp, x0, y0, wx1, wy1);
                    const int lx = x0 - (t % ntx) * TS, ly = y0 - (t / ntx) * TS;                  // in [-1, TS-1]
                    reinterpret_cast<float4*>(ids)[basep[bin[k]] + lrank[k]] =
                        make_float4(__int_as_float(rowi[k]), __int_as_float(((ly + 1) << 16) | (lx + 1)), wx1, wy1);
                }
            }
    }
}

constexpr int CHUNK = 4096;             // pairs per accumulate block

// single block: exclusive scan of `n` counts -> offsets[n+1], and of ceil(count/CHUNK) -> chunk_offsets[n+1]
__global__ void __launch_bounds__(1024) scatter_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int* __restrict__ chunk_offsets, int n) {
    __shared__ int part[1024], partc[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    int sum = 0, sumc = 0;
    for (int k = 0; k < per; ++k) { int idx = t * per + k; if (idx < n) { int c = counts[idx]; sum += c; sumc += (c + CHUNK - 1) / CHUNK; } }
    part[t] = sum; partc[t] = sumc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = t >= off ? part[t - off] : 0, vc = t >= off ? partc[t - off] : 0;
        __syncthreads();
        part[t] += v; partc[t] += vc;
        __syncthreads();
    }
    int run = part[t] - sum, runc = partc[t] - sumc;
    for (int k = 0; k < per; ++k) {
        int idx = t * per + k;
        if (idx < n) { int c = counts[idx]; offsets[idx] = run; if (chunk_offsets) chunk_offsets[idx] = runc; run += c; runc += (c + CHUNK - 1) / CHUNK; }
    }
    if (t == 1023) { offsets[n] = part[1023]; if (chunk_offsets) chunk_offsets[n] = partc[1023]; }
}

// One block per (bin, chunk of <= CHUNK pairs): the accumulation of a tile-row pair is a small GEMM on the fp32 matrix cores.
//   * wave w owns the pairs whose upper texel row is tile row w - 1 (list w; each pair is in exactly one list; list 0 = the pairs above
//     the first row, image border only).  A pair adds  wy_half * wx_x * g[c]  to (half, column x, channel c) for half in {upper, lower}
//     and x in {lx, lx+1}:  Acc[(half,x)][c] += sum_e C[(half,x)][e] * G[e][c]  with a 32 x E coefficient matrix (two rows x 16 columns)
//     that has four non-zeros per pair.  v_mfma_f32_32x32x2_f32 takes two pairs per instruction: lane (m, k) builds C[m][e_k] from the
//     pair record, lane (c, k) supplies g[e_k][c]; exact fp32 products and accumulation.  No read-modify-write chain, no atomics, no
//     dynamically indexed registers in the loop: per two pairs three LDS reads, ~10 VALU and one MFMA.
//   * the chunk is streamed in batches of 256 pairs: gradient rows + (cell, weights) records staged in LDS, bucketed by tile row;
//   * at the end the 16 row pairs are combined in the LDS tile (every tile row has exactly one "lower" and one "upper" owner) and the
//     16 x 16 x 32 tile is flushed with one atomic per touched cell channel.
// History of this scatter on MI355X (1.57 M samples x 3 planes x 4 corners x 32 channels): global float atomics 30 ms, LDS float
// atomics 4.0 ms (~0.3 lane-op/clk/CU), one-thread-per-cell owner-computes 1.7 ms, half-wave row owner with LDS read-modify-write
// 0.82 ms (a chain of ~5 dependent LDS round trips per pair and row), register accumulators selected by a scalar switch 0.48 ms /
// by indexed-VGPR moves 0.40 ms (0.14 ms of it the row gather), this form: see DESIGN.md.
constexpr int ACC_THREADS = 1024;
constexpr int ACC_BATCH = 256;
constexpr int TROWS = TS + 1;
static_assert(TROWS == 16 && ACC_THREADS == 64 * TROWS, "one wave per list, two rows x 16 columns = the 32 rows of the MFMA tile");
typedef float acc16_t __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(ACC_THREADS) scatter_accum_kernel(const float* __restrict__ df, const float4* __restrict__ pos,
                                                                    const int* __restrict__ offsets, const int* __restrict__ chunk_offsets,
                                                                    const int* __restrict__ ids, float* __restrict__ d_planes, float cs, int Hp, int Wp,
                                                                    int ldp, int ntx, int nty, int nb) {
    constexpr int STAGE_FLOATS = ACC_BATCH * FC + ACC_BATCH * 4;          // gradient rows + pair records
    constexpr int TILE_FLOATS = TROWS * TROWS * FC;
    __shared__ __attribute__((aligned(16))) float sbuf[STAGE_FLOATS > TILE_FLOATS ? STAGE_FLOATS : TILE_FLOATS];
    __shared__ int cnt[TROWS];                                           // list k holds the pairs with ly == k - 1
    __shared__ unsigned short lists[TROWS * ACC_BATCH];
    __shared__ int sbin;
    float* dfb = sbuf;
    float4* meta = reinterpret_cast<float4*>(sbuf + ACC_BATCH * FC);     // (lx, ly) as int bits, wx1, wy1
    float* tile = sbuf;                                                  // after the last batch
    const int tid = threadIdx.x;
    if (tid == 0) {
        int lo = 0, hi = nb;
        const int me = blockIdx.x;
        if (me >= chunk_offsets[nb]) lo = -1;
        else while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (chunk_offsets[mid] <= me) lo = mid; else hi = mid; }
        sbin = lo;
    }
    if (tid < TROWS) cnt[tid] = 0;
    __syncthreads();
    const int bin = sbin;
    if (bin < 0) return;
    const int chunk = blockIdx.x - chunk_offsets[bin];
    const int beg = offsets[bin] + chunk * CHUNK;
    const int end = min(offsets[bin + 1], beg + CHUNK);
    const int ntile = ntx * nty;
    const int n = bin / (3 * ntile);
    const int pl = (bin / ntile) % 3;
    const int t = bin % ntile;
    const int ty0 = (t / ntx) * TS, tx0 = (t % ntx) * TS;
    const int wave = tid >> 6, lane = tid & 63, c = lane & 31, kk = lane >> 5;
    const int mx = c & 15, mhalf = c >> 4;                               // this lane's row of the coefficient matrix: (half, column)

    acc16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // software pipeline: the (id -> position, id -> gradient row) loads of batch b+1 are in flight while batch b is accumulated
    static_assert(ACC_BATCH * (FC / 4) == 2 * ACC_THREADS, "two float4 of the gradient rows per thread");
    float4 r_meta = make_float4(0, 0, 0, 0), r_df0, r_df1;
    int r_ly = 0;
    // (batches are 256 CONSECUTIVE pairs of the chunk: taking every nbat-th pair instead, or placing the pairs of a bin in a pseudo-random
    //  order, evens out the 16 lists further -- the largest list of a batch holds ~30 of 256 pairs as it is -- but was measured slower,
    //  426 / 608 vs 389 us: the gather of the 128-byte gradient rows loses its locality)
    auto pair_at = [&](int b0, int i) { return b0 + i; };
    auto count_of = [&](int b0) { return min(ACC_BATCH, end - b0); };
    auto fetch = [&](int b0) {
        const int nbatch = count_of(b0);
        if (tid < nbatch) {
            const int row = ids[pair_at(b0, tid)] >> 2;
            int x0, y0; float wx1, wy1;
            plane_cell(pos[row], pl, cs, Hp, Wp, x0, y0, wx1, wy1);
            r_ly = y0 - ty0;                                             // in [-1, TS-1]
            r_meta = make_float4(__int_as_float(x0 - tx0), __int_as_float(r_ly), wx1, wy1);
        }
        const int j0 = tid >> 3, q = tid & 7;                           // rows j0 and j0 + 128
        r_df0 = j0 < nbatch ? reinterpret_cast<const float4*>(df + (int64_t)(ids[pair_at(b0, j0)] >> 2) * FC)[q] : make_float4(0, 0, 0, 0);
        r_df1 = j0 + 128 < nbatch ? reinterpret_cast<const float4*>(df + (int64_t)(ids[pair_at(b0, j0 + 128)] >> 2) * FC)[q] : make_float4(0, 0, 0, 0);
    };
    auto commit = [&](int b0) {
        const int nbatch = count_of(b0);
        if (tid < nbatch) {
            meta[tid] = r_meta;
            const int k = r_ly + 1;
            lists[k * ACC_BATCH + atomicAdd(&cnt[k], 1)] = (unsigned short)tid;
        }
        reinterpret_cast<float4*>(dfb)[tid] = r_df0;
        reinterpret_cast<float4*>(dfb)[tid + ACC_THREADS] = r_df1;
    };
    if (beg < end) { fetch(beg); commit(beg); }
    __syncthreads();
    for (int b0 = beg; b0 < end; b0 += ACC_BATCH) {
        const bool more = b0 + ACC_BATCH < end;
        if (more) fetch(b0 + ACC_BATCH);
        {
            const int m = cnt[wave];
            const unsigned short* lst = lists + wave * ACC_BATCH;
            // lanes 0-31 take pair e, lanes 32-63 pair e + 1; four MFMA steps per trip with all of their LDS reads issued up front (the
            // list is a pointer chase: entry -> record -> gradient row; unpipelined it costs ~5x the MFMA time when one wave holds a
            // whole batch, which happens: consecutive samples of a ray fall into the same texel row)
            for (int e = 0; e < m; e += 8) {
                int j[4];
                float4 mt[4];
                float g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) j[q] = lst[min(e + 2 * q + kk, ACC_BATCH - 1)] & (ACC_BATCH - 1);      // stale entries past m: masked below
#pragma unroll
                for (int q = 0; q < 4; ++q) { mt[q] = meta[j[q]]; g[q] = dfb[j[q] * FC + c]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool live = e + 2 * q + kk < m;
                    const int lx = __float_as_int(mt[q].x);
                    const float wx = mx == lx ? 1.f - mt[q].z : (mx == lx + 1 ? mt[q].z : 0.f);
                    const float wy = mhalf ? mt[q].w : 1.f - mt[q].w;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? wx * wy : 0.f, live ? g[q] : 0.f, acc, 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (tid < TROWS) cnt[tid] = 0;
        __syncthreads();
        if (more) commit(b0 + ACC_BATCH);
        __syncthreads();
    }
    // Combine the row pairs in the LDS tile.  Accumulator element r of a lane is coefficient row m = (r&3) + 8 (r>>2) + 4 (lane>>5), i.e.
    // (half, column) = (m >> 4, m & 15), for channel lane & 31.  List `wave` covers tile rows wave - 1 (upper) and wave (lower).
    // Lower halves first (rows 0..15, one owner each: plain stores), then the upper halves (rows 0..14: one read-modify-write owner each).
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (m >= 16) tile[(wave * TROWS + (m & 15)) * FC + c] = acc[r];
    }
    __syncthreads();
    if (wave >= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (m < 16) tile[((wave - 1) * TROWS + m) * FC + c] += acc[r];
        }
    }
    __syncthreads();
    for (int i = tid; i < TILE_FLOATS; i += ACC_THREADS) {
        const float v = tile[i];
        if (v != 0.f) {
            const int cell = i / FC, ch = i - cell * FC;
            const int yy = ty0 + cell / TROWS, xx = tx0 + cell % TROWS;
            if (yy < Hp && xx < Wp) eg3d_acc(d_planes + (int64_t)n * Hp * Wp * ldp + ((int64_t)yy * Wp + xx) * ldp + pl * FC + ch, v);
        }
    }
}

// ---- the same accumulation with every global -> LDS movement done by LDS-DMA, two batches ahead ---------------------------------------
// scatter_accum_kernel above has ONE batch of gradient rows in flight per block (in registers) behind a dependent chain
// ids -> (position, gradient row): 70 VGPRs x 16 waves = one block per CU, MfmaUtil 18 %, 2.3 TB/s of 128-byte row gathers.  Here
//   * the binning pass writes 16-byte pair records (row, cell, weights) in list order: the record stream of a chunk is contiguous and
//     goes to LDS with one dword-DMA instruction per wave and batch, three batches ahead (M);
//   * the gradient rows of batch b+2 are gathered by two 16-byte-DMA instructions per wave (eight lanes per 128-byte row, row ids read
//     from the records of batch b+2 in LDS) while batch b is accumulated (R): two batches of rows in flight per block, no registers;
//   * issue order per iteration is M(b+3), R(b+2), so `s_waitcnt vmcnt(2)` at the top of iteration b = "R(b) and M(b+2) have landed";
//   * WAVES x 64 threads, 16 / WAVES texel-row lists per wave; with 8 waves and 128-pair batches 2 blocks share a CU.
__device__ __forceinline__ void glds_b32(__amdgpu_buffer_rsrc_t rs, unsigned lds_byte, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 4, voff, 0, 0, 0);
}
__device__ __forceinline__ void glds_b128(__amdgpu_buffer_rsrc_t rs, unsigned lds_byte, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 16, voff, 0, 0, 0);
}

// workgroup barrier without the fence of __syncthreads() (that fence waits for every outstanding LDS-DMA: vmcnt(0)); LDS writes of this
// wave are complete (lgkmcnt) before it arrives
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS atomic add by address (the compiler orders atomicAdd() on __shared__ behind pending LDS-DMA with vmcnt(0))
__device__ __forceinline__ int lds_add_rtn(unsigned lds_byte, int v) {
    int old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(lds_byte), "v"(v) : "memory");
    return old;
}

template <int WAVES, int BATCH>
__global__ void __launch_bounds__(WAVES * 64) scatter_accum2_kernel(const float* __restrict__ df, int64_t S, const int* __restrict__ offsets,
                                                                     const int* __restrict__ chunk_offsets, const float4* __restrict__ recs, int64_t nrec,
                                                                     float* __restrict__ d_planes, int Hp, int Wp, int ldp, int ntx, int nty, int nb) {
    constexpr int NT = WAVES * 64;
    constexpr int LPW = TROWS / WAVES;                    // lists (upper texel rows) per wave
    constexpr int NRB = 3, NMB = 4;                       // row / record buffers in flight
    constexpr int ROWB = BATCH * FC * 4, RECB = BATCH * 16;
    static_assert(TROWS % WAVES == 0 && BATCH * 16 == NT * 4 && BATCH * FC * 4 == NT * 2 * 16, "one record dword and two row quads per thread and batch");
    constexpr int TILE_FLOATS = TROWS * TROWS * FC;
    static_assert(NRB * ROWB >= TILE_FLOATS * 4, "the flush tile reuses the row buffers");
    __shared__ __attribute__((aligned(16))) char sm[NRB * ROWB + NMB * RECB];
    __shared__ int cnt[2][TROWS];
    __shared__ unsigned short lists[TROWS * BATCH];
    __shared__ int sbin;
    float* tile = reinterpret_cast<float*>(sm);
    const unsigned lds0 = (unsigned)(uintptr_t)sm;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int lo = 0, hi = nb;
        const int me = blockIdx.x;
        if (me >= chunk_offsets[nb]) lo = -1;
        else while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (chunk_offsets[mid] <= me) lo = mid; else hi = mid; }
        sbin = lo;
    }
    if (tid < 2 * TROWS) (&cnt[0][0])[tid] = 0;
    __syncthreads();
    const int bin = sbin;
    if (bin < 0) return;
    const int chunk = blockIdx.x - chunk_offsets[bin];
    const int beg = offsets[bin] + chunk * CHUNK;
    const int end = min(offsets[bin + 1], beg + CHUNK);
    const int ntile = ntx * nty;
    const int n = bin / (3 * ntile);
    const int pl = (bin / ntile) % 3;
    const int t = bin % ntile;
    const int ty0 = (t / ntx) * TS, tx0 = (t % ntx) * TS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, kk = lane >> 5;
    const int mx = c & 15, mhalf = c >> 4;                               // this lane's row of the coefficient matrix: (half, column)
    const int nbat = (end - beg + BATCH - 1) / BATCH;

    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 0, (int)(nrec * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(df), 0, (int)(S * FC * 4), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    // M(b): thread tid moves dword tid of the batch's record stream; R(b): quad q of rows j0 and j0 + BATCH / 2
    auto issue_M = [&](int b) {
        const int i = beg + b * BATCH + (tid >> 2);
        glds_b32(rrs, lds0 + NRB * ROWB + (b % NMB) * RECB + wave * 256, (b < nbat && i < end) ? (unsigned)(i * 16 + (tid & 3) * 4) : OOB);
    };
    auto issue_R = [&](int b) {
        const float4* mb = reinterpret_cast<const float4*>(sm + NRB * ROWB + (b % NMB) * RECB);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = h * (BATCH / 2) + (tid >> 3);
            const bool ok = b < nbat && beg + b * BATCH + j < end;
            const unsigned row = (unsigned)__float_as_int(mb[j].x);
            glds_b128(drs, lds0 + (b % NRB) * ROWB + h * (ROWB / 2) + wave * 1024, ok ? row * (FC * 4) + (tid & 7) * 16 : OOB);
        }
    };

    acc16_t acc[LPW];
#pragma unroll
    for (int l = 0; l < LPW; ++l)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[l][i] = 0.f;

    // prologue -> outstanding at the top of iteration 0: M(2), R(1) x 2 (everything older has landed)
    issue_M(0); issue_M(1);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    lds_barrier();
    issue_R(0); issue_M(2);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    lds_barrier();
    issue_R(1);
    // (order is now R(0) x 2, M(2), R(1) x 2: vmcnt(2) below = R(0) and M(2) done)
    for (int b = 0; b < nbat; ++b) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        lds_barrier();
        issue_M(b + 3);
        issue_R(b + 2);
        const float4* meta = reinterpret_cast<const float4*>(sm + NRB * ROWB + (b % NMB) * RECB);
        const float* dfb = reinterpret_cast<const float*>(sm + (b % NRB) * ROWB);
        int* cn = cnt[b & 1];
        if (tid < TROWS) cnt[(b + 1) & 1][tid] = 0;
        if (tid < BATCH && beg + b * BATCH + tid < end) {
            const int k = __float_as_int(meta[tid].y) >> 16;              // ly + 1
            lists[k * BATCH + lds_add_rtn((unsigned)(uintptr_t)&cn[k], 1)] = (unsigned short)tid;
        }
        lds_barrier();
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
            const int li = wave * LPW + l;
            const int m = cn[li];
            const unsigned short* lst = lists + li * BATCH;
            for (int e = 0; e < m; e += 8) {
                int j[4];
                float4 mt[4];
                float g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) j[q] = lst[min(e + 2 * q + kk, BATCH - 1)] & (BATCH - 1);      // stale entries past m: masked below
#pragma unroll
                for (int q = 0; q < 4; ++q) { mt[q] = meta[j[q]]; g[q] = dfb[j[q] * FC + c]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool live = e + 2 * q + kk < m;
                    const int lx = (__float_as_int(mt[q].y) & 0xffff) - 1;
                    const float wx = mx == lx ? 1.f - mt[q].z : (mx == lx + 1 ? mt[q].z : 0.f);
                    const float wy = mhalf ? mt[q].w : 1.f - mt[q].w;
                    acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? wx * wy : 0.f, live ? g[q] : 0.f, acc[l], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the masked tail DMAs still write zeros: drain before the buffers become the tile
    __syncthreads();
    // combine the row pairs (see scatter_accum_kernel): lower halves first, then the upper halves
#pragma unroll
    for (int l = 0; l < LPW; ++l) {
        const int li = wave * LPW + l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (m >= 16) tile[(li * TROWS + (m & 15)) * FC + c] = acc[l][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < LPW; ++l) {
        const int li = wave * LPW + l;
        if (li >= 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m < 16) tile[((li - 1) * TROWS + m) * FC + c] += acc[l][r];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < TILE_FLOATS; i += NT) {
        const float v = tile[i];
        if (v != 0.f) {
            const int cell = i / FC, ch = i - cell * FC;
            const int yy = ty0 + cell / TROWS, xx = tx0 + cell % TROWS;
            if (yy < Hp && xx < Wp) eg3d_acc(d_planes + (int64_t)n * Hp * Wp * ldp + ((int64_t)yy * Wp + xx) * ldp + pl * FC + ch, v);
        }
    }
}

// ---- row-keyed variant (default): the pairs are sorted by (tile, upper texel row); every (tile, row) list is streamed by ONE wave ---------
// The batch lists above are rebuilt per 256 consecutive pairs of a tile, and consecutive pairs come from neighbouring rays: on the plane
// the camera faces, a ray's 96 samples fall into one texel and a strip of rays into one texel ROW -- one or two of the 16 waves do the work
// of a batch while the others wait at its barriers (MfmaUtil 18 %, 384 us; with the DMA pipeline above 311 us).  Sorting by (tile, row) in
// the binning passes gives contiguous record lists that need no batches, no per-batch lists and no barriers:
//   * binning (scatter_bin16_kernel): same two passes, 16 x the bins (3 x 18 x 18 x 16 = 15.5 K for one image: 62 KB of LDS histogram),
//     flushed sparsely -- the first arrival of a bin in a block owns its global atomic -- instead of by a sweep over all bins;
//   * accumulate (scatter_accum16p_kernel): see there.
// Measured per launch (1.57 M samples, 4.5 M pairs): zero 5 + count 32 + scan 9 + place 72 + accumulate 138 = 256 us against 470 us for
// the round-2 chain (zero 5 + 36 + 5 + 44 + 384).
// (the 62 KB histogram allows two blocks per CU: 1024-thread blocks keep the CU's wave slots full)
// What the two passes cost is the GLOBAL atomics of the flush (one per bin a block touched): a block of consecutive rows = ~57 whole rays
// crosses every depth tile of the two side planes, ~2 K bins per block, 0.6 M atomics per pass (77 us against 24 us with the flush
// disabled).  With the ray structure known (rows_per_ray, rays per image row) a block takes a BRICK instead -- 16 x 16 rays x SL
// consecutive rows of each -- whose footprint on all three planes is compact: ~4 x fewer bins touched.  Any bijection item -> row is
// correct; without the hint the blocks fall back to consecutive rows.
constexpr int BIN16_THREADS = 1024;
constexpr int BIN16_ITEMS = 18;         // (sample, plane) pairs per thread: 256 rays x 24 rows x 3 planes = 1024 x 18

template <int PASS, int SL>             // SL: rows per ray and brick (0 = consecutive rows)
__global__ void __launch_bounds__(BIN16_THREADS) scatter_bin16_kernel(const float4* __restrict__ pos, int64_t S, float cs, int Hp, int Wp, int ntx, int nty,
                                                            int nb16, int* __restrict__ counts, const int* __restrict__ offsets, int* __restrict__ fill,
                                                            float4* __restrict__ recs, int ray_w, int rpr) {
    extern __shared__ int hist[];       // [nb16] local counts; pass 1: replaced by the reserved base once the bin's owner has it
    for (int i = threadIdx.x; i < nb16; i += BIN16_THREADS) hist[i] = 0;
    __syncthreads();
    int bin[BIN16_ITEMS], lrank[BIN16_ITEMS], rowi[BIN16_ITEMS];
    const int ntile = ntx * nty;
    int brick_row0 = 0;                 // SL > 0: row of (patch ray 0, first row of the slab)
    if (SL > 0) {
        const int nslab = rpr / SL, slab = blockIdx.x % nslab, pb = blockIdx.x / nslab, ppr = ray_w / 16;
        brick_row0 = ((pb / ppr) * 16 * ray_w + (pb % ppr) * 16) * rpr + slab * SL;
    }
    // Memory operations in batches of BIN16_BATCH items: as one loop over the items -- load the position, use it, next item -- a thread made 18 memory
    // round trips in sequence in each of its two loops, and pass 1's bin owners another 18 RETURNING atomics one after the other (every
    // load / atomic of the kernel was followed by a full wait in the ISA).  Here the positions of a batch are requested before the first is used, and
    // all of a thread's reservations are in flight together.
    typedef float b16_f4 __attribute__((ext_vector_type(4)));
    constexpr int BIN16_BATCH = 6;
    static_assert(BIN16_ITEMS % BIN16_BATCH == 0, "batches");
#pragma unroll
    for (int kb = 0; kb < BIN16_ITEMS; kb += BIN16_BATCH) {
        b16_f4 psv[BIN16_BATCH];
        int plv[BIN16_BATCH];
#pragma unroll
        for (int j = 0; j < BIN16_BATCH; ++j) {
            const int k = kb + j;
            bin[k] = -1;
            int64_t row = -1; int pl = 0;
            if (SL > 0) {
                const int u = threadIdx.x + k * BIN16_THREADS;
                if (u < 256 * SL * 3) {
                    pl = u % 3;
                    const int v = u / 3, sidx = v % SL, r = v / SL;
                    row = brick_row0 + ((r >> 4) * ray_w + (r & 15)) * rpr + sidx;
                }
            } else {
                const int64_t i = ((int64_t)blockIdx.x * BIN16_ITEMS + k) * BIN16_THREADS + threadIdx.x;
                if (i < S * 3) { row = i / 3; pl = (int)(i - row * 3); }
            }
            rowi[k] = (int)row;
            plv[j] = pl;
            psv[j] = reinterpret_cast<const b16_f4*>(pos)[row >= 0 ? row : 0];          // (a valid address either way: no load under a branch)
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < BIN16_BATCH; ++j) {
            const int k = kb + j;
            const float4 ps = make_float4(psv[j].x, psv[j].y, psv[j].z, psv[j].w);
            if (rowi[k] < 0 || isnan(ps.x)) continue;
            int x0, y0; float wx1, wy1;
            if (!plane_cell(ps, plv[j], cs, Hp, Wp, x0, y0, wx1, wy1)) continue;
            const int t = tile_of(x0, y0, ntx, nty);
            bin[k] = (plv[j] * ntile + t) * TROWS + (y0 - (t / ntx) * TS + 1);
            lrank[k] = atomicAdd(&hist[bin[k]], 1);
        }
    }
    __syncthreads();
    if (PASS == 0) {
#pragma unroll
        for (int k = 0; k < BIN16_ITEMS; ++k)
            if (bin[k] >= 0 && lrank[k] == 0) atomicAdd(counts + bin[k], hist[bin[k]]);
        return;
    }
    {
        int got[BIN16_ITEMS], off[BIN16_ITEMS];
#pragma unroll
        for (int k = 0; k < BIN16_ITEMS; ++k) {
            got[k] = 0; off[k] = 0;
            if (bin[k] >= 0 && lrank[k] == 0) { got[k] = atomicAdd(fill + bin[k], hist[bin[k]]); off[k] = offsets[bin[k]]; }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < BIN16_ITEMS; ++k)
            if (bin[k] >= 0 && lrank[k] == 0) hist[bin[k]] = off[k] + got[k];
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < BIN16_ITEMS; kb += BIN16_BATCH) {
        b16_f4 psv[BIN16_BATCH];
#pragma unroll
        for (int j = 0; j < BIN16_BATCH; ++j) psv[j] = reinterpret_cast<const b16_f4*>(pos)[bin[kb + j] >= 0 ? rowi[kb + j] : 0];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < BIN16_BATCH; ++j) {
            const int k = kb + j;
            if (bin[k] >= 0) {
                const int t = (bin[k] / TROWS) % ntile, pl = bin[k] / (TROWS * ntile);
                int x0, y0; float wx1, wy1;
                plane_cell(make_float4(psv[j].x, psv[j].y, psv[j].z, psv[j].w), pl, cs, Hp, Wp, x0, y0, wx1, wy1);
                const int lx = x0 - (t % ntx) * TS;                                      // in [-1, TS-1]
                recs[hist[bin[k]] + lrank[k]] = make_float4(__int_as_float(rowi[k] | ((lx + 1) << 27)), wx1, wy1, 0.f);
            }
        }
    }
}

// single block: exclusive scan of n <= 16384 counts -> offsets[n + 1]; the counts pass through LDS so that global accesses stay coalesced
__global__ void __launch_bounds__(1024) scatter_scan16_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int n) {
    __shared__ int v[16384 + 512];
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < n; i += 1024) v[i + (i >> 5)] = counts[i];
    __syncthreads();
    const int per = (n + 1023) / 1024;
    int sum = 0;
    for (int k = 0; k < per; ++k) { const int i = t * per + k; if (i < n) sum += v[i + (i >> 5)]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int k = 0; k < per; ++k) { const int i = t * per + k; if (i < n) { const int c = v[i + (i >> 5)]; v[i + (i >> 5)] = run; run += c; } }
    if (t == 1023) offsets[n] = run;
    __syncthreads();
    for (int i = t; i < n; i += 1024) offsets[i] = v[i + (i >> 5)];
}

#if EG3D_DET
// Deterministic build: the place pass above hands out list positions with atomics, so the ORDER of a list's records differs from run to run,
// and scatter_accum16p_kernel sums a list in record order.  One block per list sorts its records by gradient row (unique within a list:
// a (sample, plane) pair has exactly one list) -- after that the list is a function of the inputs alone.  Bitonic network in its
// ascending-only ("flip") form, which needs no padding: a partner index beyond the list is skipped.  Lists of up to 4096 records are
// sorted in LDS, longer ones in place in global memory.
constexpr int SORT16_LDS = 4096;
__global__ void __launch_bounds__(256) scatter_sort16_kernel(const int* __restrict__ offsets, float4* __restrict__ recs) {
    __shared__ __attribute__((aligned(16))) float4 sm[SORT16_LDS];
    const int beg = offsets[blockIdx.x], n = offsets[blockIdx.x + 1] - beg;
    if (n < 2) return;
    float4* g = recs + beg;
    float4* a = n <= SORT16_LDS ? sm : g;
    if (n <= SORT16_LDS) {
        for (int i = threadIdx.x; i < n; i += 256) sm[i] = g[i];
    }
    __syncthreads();
    auto key = [](const float4& r) { return __float_as_int(r.x) & 0x7ffffff; };
    auto cmpx = [&](int i, int l) {
        if (l > i && l < n) {
            const float4 u = a[i], v = a[l];
            if (key(v) < key(u)) { a[i] = v; a[l] = u; }
        }
    };
    for (int k = 2; (k >> 1) < n; k <<= 1) {
        for (int i = threadIdx.x; i < n; i += 256) cmpx(i, i ^ (k - 1));
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += 256) cmpx(i, i ^ j);
            __syncthreads();
        }
    }
    if (n <= SORT16_LDS)
        for (int i = threadIdx.x; i < n; i += 256) g[i] = sm[i];
}
#endif

// ---- accumulate: persistent waves, one (tile, row) list at a time -------------------------------------------------------------------
//   * A super-step = 64 pairs: lane i loads record i (coalesced 16 bytes).  The gradient row of a pair goes straight from global memory
//     into the B register of its MFMA (two 128-byte rows per load instruction; row ids by `ds_bpermute` from the record registers).
//     Register g[q] is reloaded for the next super-step right after MFMA q consumed it: 32 loads = 8 KB in flight per wave behind the
//     in-order vmcnt, no LDS staging of the rows.
//   * The 32 x 2 coefficient matrix of an MFMA has eight non-zeros (two pairs x four corners).  Computed lane by lane it costs ~12 VALU
//     instructions per MFMA for 64 values of which 56 are zero (191 us as one block per tile, issue-bound).  Instead the non-zeros of
//     EIGHT MFMAs (16 pairs x 4 corners = 64 values) come from one instruction sequence with lane = (pair, corner), are scattered into a
//     wave-private, otherwise-zero LDS image of the eight A operands, read back densely (one ds_read per MFMA) and cleared again by a
//     second masked store.  LDS operations of a wave complete in order: no barrier anywhere in the kernel.  (150 us per tile-block.)
//   * Scheduling: a block per tile lasts as long as its longest list (1.2 x the mean) and 639 busy tiles on 256 CUs are 2.5 -> 3 rounds.
//     Here wave w of the grid takes lists w, w + 4096, ... -- one list of each plane -- and adds its 2 x 16 x 32 sums straight to the plane
//     with 16 coalesced float atomics per lane (a texel row receives the lower half of one list and the upper half of the next): 138 us.
//     Handing lists out dynamically from a global counter was slower (261 us one list per grab: the same-address atomic; 270 / 532 us
//     with 4 / 16 lists per grab: ~9.6 K busy lists are only 2.3 per wave, larger units leave waves idle).  The matrix pipe needs 59 us
//     for the 2.25 M v_mfma_f32_32x32x2_f32; variants that read one row for every pair, use two accumulators or skip the flush run within
//     2 % of the same time -- what is left is the imbalance of 2.3 lists per wave.
constexpr int ACCP_WAVES = 4;
__global__ void __launch_bounds__(ACCP_WAVES * 64) scatter_accum16p_kernel(const float* __restrict__ df, int64_t S, const int* __restrict__ offsets,
                                                                            const float4* __restrict__ recs, float* __restrict__ d_planes, int Hp, int Wp,
                                                                            int ldp, int ntx, int nty, int split) {
    __shared__ float amat[ACCP_WAVES * 8 * 64];       // per wave: the A operands of eight MFMAs
    __shared__ __attribute__((aligned(16))) float4 recm[ACCP_WAVES * 128];      // per wave: the records of two super-steps
    __shared__ __attribute__((aligned(16))) int keym[ACCP_WAVES * 128];         // ... and their row ids, [group][lane half][MFMA]
    const int tid = threadIdx.x;
    const int ntile = ntx * nty, nlist = 3 * ntile * TROWS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, kk = lane >> 5;
    float* abuf = amat + wave * (8 * 64);
    float4* recl = recm + wave * 128;
    int* keyl = keym + wave * 128;
    for (int i = lane; i < 8 * 64; i += 64) abuf[i] = 0.f;
    const int pj = lane >> 2, half = (lane >> 1) & 1, dx = lane & 1;          // this lane's (pair, corner) of a 16-pair group
    const int a_slot = (pj >> 1) * 64 + (pj & 1) * 32 + half * 16;          // + column
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(df), 0, (int)(S * FC * 4), 0x00020000);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nwave = gridDim.x * ACCP_WAVES;
    for (int ui = blockIdx.x * ACCP_WAVES + wave; ui < nlist * split; ui += nwave) {
        const int li = ui / split, part = ui - li * split;
        const int lbeg = __builtin_amdgcn_readfirstlane(offsets[li]), llen = __builtin_amdgcn_readfirstlane(offsets[li + 1]) - lbeg;
        const int beg = lbeg + (int)((int64_t)llen * part / split), end = lbeg + (int)((int64_t)llen * (part + 1) / split);
        if (beg >= end) continue;
        acc16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        auto load_recs = [&](int p0) { return p0 + lane < end ? recs[p0 + lane] : zero4; };      // an all-zero record has coefficient 0 everywhere
        // records of a super-step live in LDS (wave-private, two parities): lane (pair, corner) re-reads its pair's record as one
        // broadcast ds_read_b128, and the row ids are stored de-interleaved so that the eight ids a lane needs for a group of MFMAs
        // are two ds_read_b128 -- the cross-lane reads were 11 ds_bpermute per group before
        auto stash = [&](const float4& R, int par) {
            recl[par * 64 + lane] = R;
            keyl[par * 64 + (lane & 48) + (lane & 1) * 8 + ((lane & 15) >> 1)] = __float_as_int(R.x);
        };
        unsigned g[32];
        {
            const float4 R0 = load_recs(beg), R1 = load_recs(beg + 64);
            stash(R0, 0); stash(R1, 1);
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int4 k0 = *reinterpret_cast<const int4*>(keyl + gq * 16 + kk * 8), k1 = *reinterpret_cast<const int4*>(keyl + gq * 16 + kk * 8 + 4);
            const int nk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) g[gq * 8 + q] = __builtin_amdgcn_raw_buffer_load_b32(drs, (nk[q] & 0x7ffffff) * (FC * 4) + c * 4, 0, 0);
        }
        int par = 0;
        for (int p0 = beg; p0 < end; p0 += 64, par ^= 1) {
            const float4 Rnn = load_recs(p0 + 128);
            const int npair = min(64, end - p0);
            const bool more = p0 + 64 < end;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (gq * 16 >= npair && !more) break;
                const float4 R = recl[par * 64 + gq * 16 + pj];
                const int* kp = keyl + (par ^ 1) * 64 + gq * 16 + kk * 8;
                const int4 k0 = *reinterpret_cast<const int4*>(kp), k1 = *reinterpret_cast<const int4*>(kp + 4);
                const int nk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
                const int col = (int)((unsigned)__float_as_int(R.x) >> 27) - 1 + dx;     // -1: the corner left of the tile (no owner here)
                const float val = (dx ? R.y : 1.f - R.y) * (half ? R.z : 1.f - R.z);
                if (col >= 0) abuf[a_slot + col] = val;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int Q = gq * 8 + q;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[q * 64 + lane], __uint_as_float(g[Q]), acc, 0, 0, 0);
                    g[Q] = __builtin_amdgcn_raw_buffer_load_b32(drs, (nk[q] & 0x7ffffff) * (FC * 4) + c * 4, 0, 0);
                }
                if (col >= 0) abuf[a_slot + col] = 0.f;
            }
            stash(Rnn, par);                     // super-step p0 + 128 reuses the parity that just finished
        }
        // accumulator element r of a lane = (half, column) (m >> 4, m & 15), m = (r&3) + 8 (r>>2) + 4 kk, channel c
        const int lrow = li % TROWS, t = (li / TROWS) % ntile, pl = li / (TROWS * ntile);
        const int yy0 = (t / ntx) * TS + lrow - 1, tx0 = (t % ntx) * TS;
        float* base = d_planes + pl * FC + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int yy = yy0 + (m >> 4), xx = tx0 + (m & 15);
            if (yy >= 0 && yy < Hp && xx < Wp && acc[r] != 0.f) eg3d_acc(base + ((int64_t)yy * Wp + xx) * ldp, acc[r]);
        }
    }
}

// ---- the same accumulation on the 16-bit matrix cores (opt-in: eg3d_render_bwd_params.df_amax; EG3D_SCATTER_F16=1 in the host package) -------
// Written on the assumption that scatter_accum16p_kernel is bound by the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: two pairs per 64-cycle
// instruction).  It is not, or not only: 128 -> 112 us.  With the gradient rows coming from eight cached rows this kernel takes 93 us, with
// no row loads at all 77 us, without its MFMAs 116 us -- the random 128-byte row reads (0.6 GB per launch) and the per-pair instruction
// stream share the time.  16 us do not pay for a second arithmetic in the renderer's backward, so the exact fp32 form stays the default.
// In the arithmetic of the convolutions -- every fp32 product as three exact fp16 products of two-piece
// operands, fp32 accumulation -- v_mfma_f32_32x32x16_f16 takes SIXTEEN pairs per 32-cycle instruction: 6 instead of 32 pipe cycles per pair.
//   B (gradient rows):  g S = h + l 2^-11,  h = rtz16, l = rne16;  S = the power of two that brings max|df_rows| of the whole launch to
//                       [2^13, 2^14) (eg3d_render_bwd_params.df_amax, one atomic per block of the sample-level kernel) -- the per-tensor
//                       scaling of the convolutions' operand images; values 2^-27 below the tensor's maximum lose relative precision
//   A (coefficients):   1024 w = hw + lw, plus the plane hw 2^-11;   w S 1024 g = hw h + lw h + (hw 2^-11) l
// A group = 16 pairs = one MFMA per product: the 64 non-zero coefficients are written by lane = (pair, corner) into three otherwise-zero
// [32 rows][16 pairs] fp16 images (one ds_read_b128 per lane and plane), the lane's eight gradient values of the group -- pairs
// 8 (lane >> 5) + j -- are split in registers.
typedef _Float16 sc_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 sc_fp16x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(ACCP_WAVES * 64) scatter_accum16h_kernel(const float* __restrict__ df, int64_t S, const int* __restrict__ offsets,
                                                                            const float4* __restrict__ recs, float* __restrict__ d_planes, int Hp, int Wp,
                                                                            int ldp, int ntx, int nty, const float* __restrict__ df_amax) {
    __shared__ __attribute__((aligned(16))) _Float16 aimg[ACCP_WAVES * 3 * 512];        // per wave: three [32][16] coefficient planes
    __shared__ __attribute__((aligned(16))) float4 recm[ACCP_WAVES * 128];              // per wave: the records of two super-steps
    __shared__ __attribute__((aligned(16))) int keym[ACCP_WAVES * 128];                 // ... and their row ids
    const int tid = threadIdx.x;
    const int ntile = ntx * nty, nlist = 3 * ntile * TROWS;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, kk = lane >> 5;
    _Float16* ai = aimg + wave * (3 * 512);
    float4* recl = recm + wave * 128;
    int* keyl = keym + wave * 128;
    for (int i = lane; i < 3 * 512; i += 64) ai[i] = (_Float16)0.f;
    const int pj = lane >> 2, half = (lane >> 1) & 1, dx = lane & 1;          // this lane's (pair, corner) of a 16-pair group
    const int a_slot = half * 16 * 16 + pj;                                  // + 16 * column   (element [row = half * 16 + column][pair])
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(df), 0, (int)(S * FC * 4), 0x00020000);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // operand scale: max|g| -> [2^13, 2^14) (1 for a zero / non-finite maximum)
    float sc = 1.f;
    {
        const float am = *df_amax;
        if (am > 0.f && am < 3.0e38f) {
            int e = (int)((__float_as_uint(am) >> 23) & 0xff) - 127;
            e = max(-100, min(100, e));
            sc = __uint_as_float((unsigned)(127 + 13 - e) << 23);
        }
    }
    const float out_mul = 1.f / (sc * 1024.f);                 // exact: powers of two
    const int nwave = gridDim.x * ACCP_WAVES;
    for (int li = blockIdx.x * ACCP_WAVES + wave; li < nlist; li += nwave) {
        const int beg = __builtin_amdgcn_readfirstlane(offsets[li]), end = __builtin_amdgcn_readfirstlane(offsets[li + 1]);
        if (beg >= end) continue;
        acc16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        auto load_recs = [&](int p0) { return p0 + lane < end ? recs[p0 + lane] : zero4; };      // an all-zero record has coefficient 0 everywhere
        auto stash = [&](const float4& R, int par) { recl[par * 64 + lane] = R; keyl[par * 64 + lane] = __float_as_int(R.x); };
        // the lane's gradient value of pair 8 kk + j of group gq of the super-step whose records sit in parity `par`
        auto issue_rows = [&](unsigned (&gr)[32], int gq, int par) {
            const int* kp = keyl + par * 64 + gq * 16 + kk * 8;
            const int4 k0 = *reinterpret_cast<const int4*>(kp), k1 = *reinterpret_cast<const int4*>(kp + 4);
            const int nk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) gr[gq * 8 + j] = __builtin_amdgcn_raw_buffer_load_b32(drs, (nk[j] & 0x7ffffff) * (FC * 4) + c * 4, 0, 0);
        };
        unsigned g[32];
        {
            const float4 R0 = load_recs(beg), R1 = load_recs(beg + 64);
            stash(R0, 0); stash(R1, 1);
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) issue_rows(g, gq, 0);
        int par = 0;
        for (int p0 = beg; p0 < end; p0 += 64, par ^= 1) {
            const float4 Rnn = load_recs(p0 + 128);
            const int npair = min(64, end - p0);
            const bool more = p0 + 64 < end;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                if (gq * 16 >= npair && !more) break;
                // coefficients of the group's 16 pairs x 4 corners
                const float4 R = recl[par * 64 + gq * 16 + pj];
                const int col = (int)((unsigned)__float_as_int(R.x) >> 27) - 1 + dx;     // -1: the corner left of the tile (no owner here)
                const float w = (dx ? R.y : 1.f - R.y) * (half ? R.z : 1.f - R.z) * 1024.f;
                const sc_fp16x2 hw2 = __builtin_amdgcn_cvt_pkrtz(w, 0.f);
                const _Float16 hw = (_Float16)hw2[0], lw = (_Float16)(w - (float)hw2[0]), gw = hw * (_Float16)0.00048828125f;
                if (col >= 0) { ai[a_slot + col * 16] = hw; ai[512 + a_slot + col * 16] = lw; ai[1024 + a_slot + col * 16] = gw; }
                // gradient values: split, then the registers are free for the next super-step's rows
                sc_f16x8 bh, bl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = __uint_as_float(g[gq * 8 + 2 * q]) * sc, a1 = __uint_as_float(g[gq * 8 + 2 * q + 1]) * sc;
                    const sc_fp16x2 hh = __builtin_amdgcn_cvt_pkrtz(a0, a1);
                    bh[2 * q] = (_Float16)hh[0]; bh[2 * q + 1] = (_Float16)hh[1];
                    bl[2 * q] = (_Float16)__builtin_amdgcn_fmed3f((a0 - (float)hh[0]) * 2048.f, -65504.f, 65504.f);
                    bl[2 * q + 1] = (_Float16)__builtin_amdgcn_fmed3f((a1 - (float)hh[1]) * 2048.f, -65504.f, 65504.f);
                }
                issue_rows(g, gq, par ^ 1);
                const sc_f16x8* ap = reinterpret_cast<const sc_f16x8*>(ai) + (lane & 31) * 2 + kk;       // row (lane & 31), pairs 8 kk .. 8 kk + 7
                const sc_f16x8 ah = ap[0], al = ap[64], ag = ap[128];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, bl, acc, 0, 0, 0);                        // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                if (col >= 0) { ai[a_slot + col * 16] = (_Float16)0.f; ai[512 + a_slot + col * 16] = (_Float16)0.f; ai[1024 + a_slot + col * 16] = (_Float16)0.f; }
            }
            stash(Rnn, par);                     // super-step p0 + 128 reuses the parity that just finished
        }
        // accumulator element r of a lane = (half, column) (m >> 4, m & 15), m = (r&3) + 8 (r>>2) + 4 kk, channel c
        const int lrow = li % TROWS, t = (li / TROWS) % ntile, pl = li / (TROWS * ntile);
        const int yy0 = (t / ntx) * TS + lrow - 1, tx0 = (t % ntx) * TS;
        float* base = d_planes + pl * FC + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int yy = yy0 + (m >> 4), xx = tx0 + (m & 15);
            const float v = acc[r] * out_mul;
            if (yy >= 0 && yy < Hp && xx < Wp && v != 0.f) eg3d_acc(base + ((int64_t)yy * Wp + xx) * ldp, v);
        }
    }
}

__global__ void render_finalize_kernel(float* depth, const float* mm, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = depth[i];
    if (isnan(v)) v = INFINITY;            // nan_to_num(nan=inf)
    if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;   // posinf/neginf -> finite max (only original infs)
    depth[i] = fminf(fmaxf(v, mm[0]), mm[1]);
}

// ---- ray generation -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ray_gen_fwd_kernel(const float* __restrict__ c2w, const float* __restrict__ K, float* __restrict__ origins,
                                                          float* __restrict__ dirs, int N, int res) {
    const int64_t total = (int64_t)N * res * res;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i / (res * res));
    const int pix = (int)(i - (int64_t)n * res * res);
    const int row = pix / res, col = pix - row * res;
    const float* M = c2w + n * 16;
    const float* Kn = K + n * 9;
    const float fx = Kn[0], sk = Kn[1], cx = Kn[2], fy = Kn[4], cy = Kn[5];
    const float inv = (float)(1.0 / res), half = (float)(0.5 / res);
    const float xc = (float)col * inv + half, yc = (float)row * inv + half;
    const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
    const float yl = (yc - cy) / fy;
    float v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float w = M[a * 4 + 0] * xl + M[a * 4 + 1] * yl + M[a * 4 + 2] + M[a * 4 + 3];
        v[a] = w - M[a * 4 + 3];
        origins[i * 3 + a] = M[a * 4 + 3];
    }
    float nrm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    dirs[i * 3 + 0] = v[0] / nrm; dirs[i * 3 + 1] = v[1] / nrm; dirs[i * 3 + 2] = v[2] / nrm;
}

// grid = (blocks, N); accumulates with atomics into pre-zeroed d_c2w [N,16], d_K [N,9]
__global__ void __launch_bounds__(256) ray_gen_bwd_kernel(const float* __restrict__ c2w, const float* __restrict__ K, const float* __restrict__ g_o,
                                                          const float* __restrict__ g_d, float* __restrict__ d_c2w, float* __restrict__ d_K, int res) {
    __shared__ float red[4][17];
    const int n = blockIdx.y;
    const float* M = c2w + n * 16;
    const float* Kn = K + n * 9;
    const float fx = Kn[0], sk = Kn[1], cx = Kn[2], fy = Kn[4], cy = Kn[5];
    const float inv = (float)(1.0 / res), half = (float)(0.5 / res);
    float acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.f;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < res * res; pix += gridDim.x * blockDim.x) {
        const int row = pix / res, col = pix - row * res;
        const int64_t i = (int64_t)n * res * res + pix;
        const float xc = (float)col * inv + half, yc = (float)row * inv + half;
        const float xl = (xc - cx + cy * sk / fy - sk * yc / fy) / fx;
        const float yl = (yc - cy) / fy;
        float v[3], g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            v[a] = M[a * 4 + 0] * xl + M[a * 4 + 1] * yl + M[a * 4 + 2];
            g[a] = g_d ? g_d[i * 3 + a] : 0.f;
        }
        float nrm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
        float dn[3] = {v[0] / nrm, v[1] / nrm, v[2] / nrm};
        float dot = dn[0] * g[0] + dn[1] * g[1] + dn[2] * g[2];
        float dxl = 0.f, dyl = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float dv = (g[a] - dn[a] * dot) / nrm;
            acc[a * 4 + 0] += dv * xl; acc[a * 4 + 1] += dv * yl; acc[a * 4 + 2] += dv;
            acc[a * 4 + 3] += g_o ? g_o[i * 3 + a] : 0.f;
            dxl += dv * M[a * 4 + 0]; dyl += dv * M[a * 4 + 1];
        }
        // intrinsics: order fx, sk, cx, fy, cy -> acc[12..16]
        acc[12] += -xl / fx * dxl;
        acc[13] += (cy / fy - yc / fy) / fx * dxl;
        acc[14] += -dxl / fx;
        acc[15] += dxl * (-cy * sk / (fy * fy) + sk * yc / (fy * fy)) / fx - (yc - cy) / (fy * fy) * dyl;
        acc[16] += (sk / fy) / fx * dxl - dyl / fy;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        float t = acc[k];
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
        if (lane == 0) red[wv][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        int k = threadIdx.x;
        if (k < 12) eg3d_acc(d_c2w + n * 16 + k, t);
        else if (d_K) {
            const int map[5] = {0, 1, 2, 4, 5};
            eg3d_acc(d_K + n * 9 + map[k - 12], t);
        }
    }
}

int check_render(const eg3d_render_params& p) {
    if (!p.planes || !p.origins || !p.dirs || !p.u1 || !p.w0 || !p.b0 || !p.w1 || !p.b1) return EG3D_ERR_INVALID;
    if (p.N <= 0 || p.R <= 0 || p.Hp <= 0 || p.Wp <= 0 || p.Dc < 2 || p.Df < 0) return EG3D_ERR_INVALID;
    if (p.C != FC || p.Hdim != HD || p.Cout != CO) return EG3D_ERR_UNSUPPORTED;
    if (p.ldp < 3 * FC || (p.ldp & 3) || (reinterpret_cast<uintptr_t>(p.planes) & 15)) return EG3D_ERR_UNSUPPORTED;
    const int D = p.Dc > p.Df ? p.Dc : p.Df;
    if (D > MAXT || D < 6) return EG3D_ERR_UNSUPPORTED;
    if (p.Df > 0 && (!p.u2 || !p.fine_depths || p.Dc < 4)) return EG3D_ERR_INVALID;
    return EG3D_OK;
}

// positions of the coarse samples for the sample-level kernel: pos_rows[0][ray][s] = (o + t d, t); x = NaN marks an absent sample
__global__ void __launch_bounds__(256) coarse_pos_kernel(const eg3d_render_params p, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nrays = (int64_t)p.N * p.R;
    if (i == 0) { p.depth_minmax[0] = INFINITY; p.depth_minmax[1] = -INFINITY; }     // first launch of the pipelined forward: the caller need not initialise it
    if (i >= nrays * D) return;
    const int64_t ray = i / D;
    const int s = (int)(i - ray * D);
    float4 ps = make_float4(NAN, 0.f, 0.f, 0.f);
    if (s < p.Dc) {
        const float t = coarse_depth(p, ray, s, p.u1[ray * p.Dc + s]);
        ps = make_float4(p.origins[ray * 3] + t * p.dirs[ray * 3], p.origins[ray * 3 + 1] + t * p.dirs[ray * 3 + 1],
                         p.origins[ray * 3 + 2] + t * p.dirs[ray * 3 + 2], t);
    }
    reinterpret_cast<float4*>(p.pos_rows)[i] = ps;
}

size_t render_smem(const eg3d_render_params& p, int mode) {
    const int D = p.Dc > p.Df ? p.Dc : p.Df;
    const int RPB = MAXT / D;
    return ((size_t)RPB * RAY_LDS_FLOATS(D) + render_red_floats(RPB, D, mode) + (size_t)RPB * CO) * sizeof(float);
}

}  // namespace

extern "C" int eg3d_render_fwd(const eg3d_render_params* pp, void* stream) {
    if (!pp) return EG3D_ERR_INVALID;
    int rc = check_render(*pp);
    if (rc) return rc;
    if (!pp->rgb || !pp->depth || !pp->wsum || !pp->depth_minmax) return EG3D_ERR_INVALID;
    const int D = pp->Dc > pp->Df ? pp->Dc : pp->Df;
    const int RPB = MAXT / D;
    eg3d_render_bwd_params bp = {};
    bp.fwd = *pp;
    const int64_t nrays = (int64_t)pp->N * pp->R;
    hipStream_t st = (hipStream_t)stream;
    if (pp->pos_rows != nullptr && pp->save_sigma != nullptr && pp->save_rgb != nullptr && pp->fine_depths != nullptr && pp->Df > 0) {
        // pipelined forward: positions -> decode (matrix cores) -> importance sampling -> decode -> compositing
        const int64_t M = nrays * D;
        hipLaunchKernelGGL(coarse_pos_kernel, dim3(eg3d_cdiv(M, 256)), dim3(256), 0, st, *pp, D);
        rc = eg3d_decode_rows_fwd(*pp, pp->pos_rows, 4, M, (int64_t)pp->R * D, pp->save_sigma, pp->save_rgb, stream, D, 2 * D, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(render_kernel<2>, dim3(eg3d_cdiv(nrays, RPB)), dim3(MAXT), render_smem(*pp, 2), st, bp);
        rc = eg3d_decode_rows_fwd(*pp, pp->pos_rows + M * 4, 4, M, (int64_t)pp->R * D, pp->save_sigma, pp->save_rgb, stream, D, 2 * D, D);
        if (rc) return rc;
        hipLaunchKernelGGL(render_kernel<3>, dim3(eg3d_cdiv(nrays, RPB)), dim3(MAXT), render_smem(*pp, 3), st, bp);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
    hipLaunchKernelGGL(render_kernel<0>, dim3(eg3d_cdiv(nrays, RPB)), dim3(MAXT), render_smem(*pp, 0), st, bp);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_render_finalize(float* depth, const float* depth_minmax, int64_t n, void* stream) {
    if (!depth || !depth_minmax || n < 0) return EG3D_ERR_INVALID;
    if (n == 0) return EG3D_OK;
    hipLaunchKernelGGL(render_finalize_kernel, dim3(eg3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, depth, depth_minmax, n);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_render_bwd(const eg3d_render_bwd_params* bp, void* stream) {
    if (!bp) return EG3D_ERR_INVALID;
    int rc = check_render(bp->fwd);
    if (rc) return rc;
    const eg3d_render_params& p = bp->fwd;
    if (!bp->d_rgb || !p.depth_minmax || !p.save_sigma || !p.save_rgb || !bp->ag_rows) return EG3D_ERR_INVALID;
    if ((bp->d_origins || bp->d_dirs) && !bp->gc_rows) return EG3D_ERR_INVALID;
    if (!bp->df_pos) return EG3D_ERR_INVALID;
    const int D = p.Dc > p.Df ? p.Dc : p.Df;
    const int RPB = MAXT / D;
    const int64_t nrays = (int64_t)p.N * p.R;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(render_kernel<1>, dim3(eg3d_cdiv(nrays, RPB)), dim3(MAXT), render_smem(p, 1), st, *bp);
    const int64_t S = nrays * 2 * D;
    EG3D_DET_SCOPE(det, stream);
    EG3D_DET_BIND(det, bp->gram_w0, HD * FC); EG3D_DET_BIND(det, bp->gram_b0, HD); EG3D_DET_BIND(det, bp->gram_w1, (1 + CO) * HD); EG3D_DET_BIND(det, bp->gram_b1, 1 + CO);
    EG3D_DET_COMMIT(det);
    if (int rc2 = eg3d_decode_rows_bwd(*bp, bp->df_pos, 0, S, (int64_t)p.R * 2 * D, 2 * D, stream)) return rc2;
    EG3D_DET_END(det);
    if (bp->d_origins || bp->d_dirs)
        hipLaunchKernelGGL(render_coord_reduce_kernel, dim3(eg3d_cdiv(nrays, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(bp->gc_rows),
                           bp->d_origins, bp->d_dirs, nrays, 2 * D);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_render_query_sizes(const eg3d_render_params* p, eg3d_render_sizes* out) {
    if (!p || !out || p->N <= 0 || p->R <= 0 || p->Dc <= 0 || p->Df < 0 || p->Cout <= 0) return EG3D_ERR_INVALID;
    const int64_t NR = (int64_t)p->N * p->R, D = p->Dc > p->Df ? p->Dc : p->Df, S = NR * 2 * D;
    out->S = S;
    out->rgb = NR * p->Cout;  out->depth = NR;  out->wsum = NR;  out->depth_minmax = 2;
    out->fine_depths = NR * p->Df;
    out->save_sigma = S;  out->save_rgb = S * p->Cout;  out->pos_rows = p->Df > 0 ? 2 * NR * D * 4 : 0;
    out->df_rows = S * FC;  out->df_pos = S * 4;  out->ag_rows = S * 2;  out->gc_rows = S * 4;
    out->dump_dpre = S * 64;  out->dump_h = S * 64;  out->dump_dout = S * (1 + p->Cout);  out->dump_feat = S * FC;
    out->feat_rows = p->Df > 0 ? S * FC : 0;
    return EG3D_OK;
}

extern "C" int64_t eg3d_triplane_scatter_workspace_ints(int64_t S, int N, int Hp, int Wp) {
    const int64_t nb = (int64_t)N * 3 * ((Hp + TS - 1) / TS) * ((Wp + TS - 1) / TS);
    const int64_t nb16 = 3 * ((Hp + TS - 1) / TS) * ((Wp + TS - 1) / TS) * (TS + 1);        // row-keyed bins of one image (3 arrays)
    return (4 * nb + 2 > 3 * nb16 + 2 ? 4 * nb + 2 : 3 * nb16 + 2) + 3 + 12 * S;            // counts, fill, offsets (+1), chunk_offsets (+1), [pad to 16 bytes] pair records (16 bytes per (sample, plane))
}

extern "C" int eg3d_triplane_scatter(const float* df_rows, const float* df_pos, int64_t S, int64_t rows_per_image, float* d_planes, int N, int Hp,
                                     int Wp, int ldp, float box_warp, int32_t* workspace, int ray_w, int rows_per_ray, const float* df_amax, void* stream) {
    if (!df_rows || !df_pos || !d_planes || !workspace || S <= 0 || N <= 0 || rows_per_image <= 0) return EG3D_ERR_INVALID;
    if (ldp < 3 * FC || 3 * S > INT32_MAX / 4) return EG3D_ERR_UNSUPPORTED;
    const int ntx = (Wp + TS - 1) / TS, nty = (Hp + TS - 1) / TS;
    const int nb = N * 3 * ntx * nty;
    if ((size_t)nb * 2 * sizeof(int) > 64 * 1024) return EG3D_ERR_UNSUPPORTED;       // LDS histogram of the binning kernels
    int* counts = workspace;
    int* fill = counts + nb;
    int* offsets = fill + nb;
    int* chunk_offsets = offsets + nb + 1;
    int* ids = chunk_offsets + nb + 1;
    ids += (4 - ((ids - workspace) & 3)) & 3;          // the records are float4 (the workspace itself is at least 16-byte aligned)
    hipStream_t st = (hipStream_t)stream;
    const float cs = 2.f / box_warp;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, d_planes, (int64_t)N * Hp * Wp * ldp); EG3D_DET_COMMIT(det);
    const int blocks = eg3d_cdiv(S * 3, 256 * BIN_ITEMS);
    const float4* pos4 = reinterpret_cast<const float4*>(df_pos);
    static const int variant0 = [] { const char* e = getenv("EG3D_SCATTER"); return e ? atoi(e) : 4; }();
    const int nb16_img = 3 * ntx * nty * TROWS;                 // row-keyed bins of ONE image
    if (variant0 == 4 && nb16_img <= 16384 && S % N == 0 && rows_per_image == S / N && rows_per_image * FC * 4 < ((int64_t)1 << 32)
        && rows_per_image < (1 << 27)) {
        // row-keyed sort + wave-streamed accumulation, image by image (bins, records and row ids are per image)
        int* counts16 = workspace;
        int* fill16 = counts16 + nb16_img;
        int* offsets16 = fill16 + nb16_img;
        int* recs_i = offsets16 + nb16_img + 1;
        recs_i += (4 - ((recs_i - workspace) & 3)) & 3;
        float4* recs = reinterpret_cast<float4*>(recs_i);
        const int64_t Si = rows_per_image;
        // brick mode: 16 x 16 rays x SL rows per block
        int SL = 0;
        if (ray_w > 0 && rows_per_ray > 0 && ray_w % 16 == 0 && Si % ((int64_t)rows_per_ray * ray_w) == 0 && (Si / ((int64_t)rows_per_ray * ray_w)) % 16 == 0)
            SL = rows_per_ray % 24 == 0 ? 24 : (rows_per_ray % 16 == 0 ? 16 : 0);
        const int blocks_i = SL ? (int)(Si / rows_per_ray / 256 * (rows_per_ray / SL)) : eg3d_cdiv(Si * 3, BIN16_THREADS * BIN16_ITEMS);
        auto bin_pass = [&](int pass, const float4* pos_n) {
#define EG3D_BIN16(P, L) hipLaunchKernelGGL((scatter_bin16_kernel<P, L>), dim3(blocks_i), dim3(BIN16_THREADS), sizeof(int) * nb16_img, st, pos_n, Si, cs, Hp, \
                                            Wp, ntx, nty, nb16_img, counts16, offsets16, fill16, recs, ray_w, rows_per_ray)
            if (pass == 0) { if (SL == 24) EG3D_BIN16(0, 24); else if (SL == 16) EG3D_BIN16(0, 16); else EG3D_BIN16(0, 0); }
            else           { if (SL == 24) EG3D_BIN16(1, 24); else if (SL == 16) EG3D_BIN16(1, 16); else EG3D_BIN16(1, 0); }
#undef EG3D_BIN16
        };
        for (int n = 0; n < N; ++n) {
            const float4* pos_n = pos4 + (int64_t)n * Si;
            eg3d_zero_words(counts16, 2 * (int64_t)nb16_img, st);
            bin_pass(0, pos_n);
            hipLaunchKernelGGL(scatter_scan16_kernel, dim3(1), dim3(1024), 0, st, counts16, offsets16, nb16_img);
            bin_pass(1, pos_n);
#if EG3D_DET
            hipLaunchKernelGGL(scatter_sort16_kernel, dim3(nb16_img), dim3(256), 0, st, offsets16, recs);
#endif
            constexpr int split = 1;
            // one block of four waves per four lists: the hardware's block dispatch does the load balancing (lists differ 0 .. 990 pairs); with
            // 1024 persistent blocks taking lists w, w + 4096, ... the launch lasted as long as its unluckiest wave (fp32: 142 -> 128 us)
            const int accb = (nb16_img + ACCP_WAVES - 1) / ACCP_WAVES;
            if (df_amax != nullptr) {
                hipLaunchKernelGGL(scatter_accum16h_kernel, dim3(accb), dim3(ACCP_WAVES * 64), 0, st, df_rows + (int64_t)n * Si * FC, Si, offsets16, recs,
                                   d_planes + (int64_t)n * Hp * Wp * ldp, Hp, Wp, ldp, ntx, nty, df_amax);
                continue;
            }
            hipLaunchKernelGGL(scatter_accum16p_kernel, dim3(accb), dim3(ACCP_WAVES * 64), 0, st, df_rows + (int64_t)n * Si * FC, Si, offsets16, recs,
                               d_planes + (int64_t)n * Hp * Wp * ldp, Hp, Wp, ldp, ntx, nty, split);
        }
        EG3D_DET_END(det);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
#if EG3D_DET
    return EG3D_ERR_UNSUPPORTED;          // the older scatter variants sum in an order their LDS cursors decide
#endif
    eg3d_zero_words(counts, 2 * (int64_t)nb, st);          // counts + fill cursors (a kernel, not a memset node: see common.h)
    hipLaunchKernelGGL(scatter_bin_kernel<0>, dim3(blocks), dim3(256), sizeof(int) * nb, st, pos4, S, rows_per_image, cs, Hp, Wp, ntx, nty, nb, counts,
                       offsets, fill, ids);
    hipLaunchKernelGGL(scatter_scan_kernel, dim3(1), dim3(1024), 0, st, counts, offsets, chunk_offsets, nb);
    const int max_chunks = (int)((3 * S + CHUNK - 1) / CHUNK) + nb;
    const int variant = variant0 == 4 ? 3 : variant0;      // 1: round-2 kernel; 2 / 3: DMA-pipelined batches (8 / 16 waves); 4 (default, above): row-keyed
    if (variant == 1 || S * FC * 4 >= ((int64_t)1 << 32) || 3 * S * 16 >= ((int64_t)1 << 32)) {
        hipLaunchKernelGGL(scatter_bin_kernel<1>, dim3(blocks), dim3(256), sizeof(int) * 2 * nb, st, pos4, S, rows_per_image, cs, Hp, Wp, ntx, nty, nb, counts,
                           offsets, fill, ids);
        hipLaunchKernelGGL(scatter_accum_kernel, dim3(max_chunks), dim3(ACC_THREADS), 0, st, df_rows, pos4, offsets, chunk_offsets, ids, d_planes, cs, Hp, Wp, ldp,
                           ntx, nty, nb);
    } else {
        hipLaunchKernelGGL(scatter_bin_kernel<2>, dim3(blocks), dim3(256), sizeof(int) * 2 * nb, st, pos4, S, rows_per_image, cs, Hp, Wp, ntx, nty, nb, counts,
                           offsets, fill, ids);
        const float4* recs = reinterpret_cast<const float4*>(ids);
        hipLaunchKernelGGL((scatter_accum2_kernel<16, 256>), dim3(max_chunks), dim3(1024), 0, st, df_rows, S, offsets, chunk_offsets, recs, 3 * S, d_planes, Hp, Wp,
                           ldp, ntx, nty, nb);
    }
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_sample_decode(const eg3d_render_params* pp, const float* coords, int64_t M, float* rgb, float* sigma, void* stream) {
    if (!pp || !coords || !rgb || !sigma || M < 0) return EG3D_ERR_INVALID;
    const eg3d_render_params& p = *pp;
    if (!p.planes || !p.w0 || !p.b0 || !p.w1 || !p.b1 || p.N <= 0) return EG3D_ERR_INVALID;
    if (p.C != FC || p.Hdim != HD || p.Cout != CO || p.ldp < 3 * FC || (p.ldp & 3)) return EG3D_ERR_UNSUPPORTED;
    if (M == 0) return EG3D_OK;
    eg3d_render_params q = p;
    q.feat_rows = nullptr;               // (a buffer of the ray renderer's row count: not for free-standing points)
    return eg3d_decode_rows_fwd(q, coords, 3, (int64_t)p.N * M, M, sigma, rgb, stream);
}

extern "C" int eg3d_ray_gen_fwd(const float* cam2world, const float* intrinsics, float* origins, float* dirs, int N, int res, void* stream) {
    if (!cam2world || !intrinsics || !origins || !dirs || N <= 0 || res <= 0) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(ray_gen_fwd_kernel, dim3(eg3d_cdiv((int64_t)N * res * res, 256)), dim3(256), 0, (hipStream_t)stream, cam2world, intrinsics,
                       origins, dirs, N, res);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_ray_gen_bwd(const float* cam2world, const float* intrinsics, const float* d_origins, const float* d_dirs, float* d_cam2world,
                                float* d_intrinsics, int N, int res, void* stream) {
    if (!cam2world || !intrinsics || !d_cam2world || N <= 0 || res <= 0) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    eg3d_zero_words(d_cam2world, 16 * (int64_t)N, st);
    if (d_intrinsics) eg3d_zero_words(d_intrinsics, 9 * (int64_t)N, st);
    int bx = std::min(eg3d_cdiv((int64_t)res * res, 256), 64);
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, d_cam2world, 16 * (int64_t)N); EG3D_DET_BIND(det, d_intrinsics, 9 * (int64_t)N); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(ray_gen_bwd_kernel, dim3(bx, N), dim3(256), 0, st, cam2world, intrinsics, d_origins, d_dirs, d_cam2world, d_intrinsics, res);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
