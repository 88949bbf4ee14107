// Sample-level kernels of the volume renderer: tri-plane gather + OSG decoder (32 -> 64 softplus -> 1+32), forward and
// backward, with the decoder on the 16-bit matrix cores at fp32-equivalent precision (three v_mfma_f32_32x32x16_f16 per product
// block on two-piece fp16 operands, the arithmetic of conv_v2.hip; fp32 accumulation).
//
// Replaces sample_from_planes/F.grid_sample (training/volumetric_rendering/renderer.py:55-66) + OSGDecoder.forward
// (training/triplane.py:124-136) and their autograd backward, for arbitrary sample positions ("rows").
//
// Mapping.  A wave processes tiles of 32 samples; the two lanes l and l+32 share sample l of the tile and each holds half of
// every 32-vector (split_idx, render_common.h).  All four GEMMs are computed transposed (result rows = output units, columns =
// samples), so the MFMA result layout of one layer *is* the B-operand layout of the next one:
//     PRE^T[64 x 32s] = W0 F^T          H = softplus(PRE)          OUT^T[32 x 32s] = W1c H^T       sigma = w1s . H  (VALU + 1 shuffle)
//     dH^T = W1c^T dOUT^T + w1s dsigma  dPRE = dH (1 - exp(-H))    dF^T [32 x 32s] = W0^T dPRE^T
// A k-step of the 16-bit MFMA takes 8 consecutive registers of the split layout per lane (K index = split_idx(8 s + j, lane >> 5)),
// the weight-side (A) operands are staged once per block in LDS in that same K order as three fp16 planes (high piece, low piece,
// high piece x 2^-11), one conflict-free ds_read_b128 each.  The activation-side operands are split in registers:
//     x = h + l 2^-11,  h = rtz16(x),  l = rne16((x - h) 2^11);      w 2^e = hw + lw;      x w 2^e = h hw + h lw + l (hw 2^-11)
// Gradient operands (dOUT, dPRE) span many decades from sample to sample, so each sample (= GEMM column) is brought into fp16 range
// by its own power of two, undone exactly on the result column.  48 MFMAs (32 cycles each) per 32 samples for the backward kernel
// instead of 128 fp32 ones (64 cycles each); 24 instead of 64 for the forward one.  The gather is done in the same split layout: the
// two lanes of a sample read the two 64-byte halves of each 128-byte texel.
#include "render_common.h"
#include "det.h"
#include <atomic>
#include <cstdlib>

using namespace eg3d_render;

namespace {

struct DecodeArgs {
    const float* planes; int N, Hp, Wp, ldp; float cs;
    const float* w0; const float* b0; const float* w1t; const float* b1;     // gains folded; w1t = [HD][1+CO]
    const float* pos;           // [M, pos_stride] (x,y,z[,depth]); x = NaN -> row skipped
    int pos_stride;             // 3 or 4 floats per row
    int64_t M;                  // rows
    int64_t rows_per_image;     // image index of row i = i / rows_per_image
    float* sigma;               // [M]
    float* rgb;                 // [M,CO]
    int seg_len, seg_stride, seg_off;   // forward outputs: logical row i is written at (i / seg_len) * seg_stride + seg_off + i % seg_len (seg_len 0 = identity)
    // backward only
    const float2* ag;           // [M] (colour weight a, dL/d sigma)
    const float* d_rgb;         // [rays,CO] incoming per-ray colour gradient
    int64_t samples_per_ray_row;// ray index of row i = i / samples_per_ray_row  (within this launch)
    int64_t ray0;               // first ray of this launch
    float* df_rows;             // [M,FC] dL/d(mean feature) / 3        (or null)
    float4* gc_rows;            // [M] (dL/d position, depth)            (or null)
    float* dump_dpre; float* dump_h; float* dump_dout; float* dump_feat;   // decoder-weight gradient operands (or null)
    float* df_amax;             // backward: max|df_rows| over the launch (one atomic per block), or null
    float* gram_w0; float* gram_b0; float* gram_w1; float* gram_b1;        // GRAM instantiation: decoder-weight gradients contracted in the kernel
    float gram_s0, gram_s1, gram_sb;
    float* feat;                // FEAT instantiations: [rows, FC] interpolated features in OUTPUT row order (written by gather_rows_kernel)
    int nt_rows;                // gather_rows_kernel: feature rows stored non-temporally (set by launch_decode when the row buffer exceeds the last-level cache)
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS fragment images of the weights: per (output block, k-step) three planes (hw, lw, hw 2^-11) of 64 lanes x 8 halfs.
// K index of slot j of lane l in k-step s: kidx(s, l >> 5, j) = split_idx(8 s + j, l >> 5).
struct Frags {
    const f16x8* a1;   // [2 ht][2 s][3][64]   W0[32 ht + (l&31)][kidx]                         (layer 1:  i = hidden, k = feature)
    const f16x8* a2;   // [4 ks][3][64]        W1[1 + (l&31)][32 (ks>>1) + kidx(ks&1)]          (layer 2:  i = colour, k = hidden)
    const f16x8* a3;   // [2 ht][2 s][3][64]   W1[1 + kidx][32 ht + (l&31)]                     (dH:       i = hidden, k = colour)
    const f16x8* a4;   // [4 ks][3][64]        W0[32 (ks>>1) + kidx(ks&1)][(l&31)]              (dF:       i = feature, k = hidden)
    const float* ws;   // [2 h][2 ht][16]   W1[0][32ht + idx(r,h)]                              (sigma row, fp32; a lane reads its 16 values as float4s)
    const float* bi0;  // [2 h][2 ht][16]   b0[32ht + idx(r,h)]
    const float* bi1;  // [2 h][16]         b1[1 + idx(r,h)]
    float inv0, inv1;  // 1 / (power-of-two range multiplier of W0, W1)
};
constexpr int FRAG_PLANES = 4 * 3 * 64;                 // f16x8 units of one GEMM's fragment image
constexpr int FRAG_BYTES_FWD = 2 * FRAG_PLANES * 16 + (64 + 64 + 32 + 16) * 4;
constexpr int FRAG_BYTES_BWD = 4 * FRAG_PLANES * 16 + (64 + 64 + 32 + 16) * 4;

// multiplier that brings `amax` to [2^11, 2^12): an exact power of two (1 for zero / non-finite input)
__device__ __forceinline__ float pow2_range_mul(float amax, int target_exp) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;          // floor(log2 amax) (subnormals: -127)
    e = max(-100, min(100, e));
    return __uint_as_float((unsigned)(127 + target_exp - e) << 23);
}

__device__ __forceinline__ void split_pieces8(const float (&v)[8], float lo_mul, f16x8& h, f16x8& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        const fp16x2_t hh = __builtin_amdgcn_cvt_pkrtz(a, b);
        const float ra = __builtin_amdgcn_fmed3f((a - (float)hh[0]) * lo_mul, -65504.f, 65504.f);
        const float rb = __builtin_amdgcn_fmed3f((b - (float)hh[1]) * lo_mul, -65504.f, 65504.f);
        h[2 * q] = (_Float16)hh[0]; h[2 * q + 1] = (_Float16)hh[1];
        l[2 * q] = (_Float16)ra; l[2 * q + 1] = (_Float16)rb;
    }
}

// B-operand fragments (high piece, low piece scaled by 2^11) of k-step s of a 16-register split-layout vector, times `mul` (a power of two).
// The sample-level kernels are VALU-bound (round 4: ~1650 vector instructions against 48 MFMAs per 32 samples in the backward kernel), so
// the split is written for the packed fp32 pipe: per PAIR of elements one v_pk_mul (range), one v_cvt_pkrtz (high pieces), one v_pk_mul
// (x 2^11) and two v_fma_mix{lo,hi}_f16 -- fma(-h, 2^11, x 2^11) with the f16 operand read in place and the result rounded ONCE to f16
// (the product and the difference are exact) -- 2.5 instructions per element instead of 6.  RANGE = false: no multiplier (features,
// hidden activations).  CLAMP: the input is held inside the fp16 range first (features are whatever the planes hold; an operand the
// caller normalised needs none).
template <bool RANGE, bool CLAMP>
__device__ __forceinline__ void act_frag(const f32x16& v, int s, float mul, f16x8& h, f16x8& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2 x = {v[8 * s + 2 * q], v[8 * s + 2 * q + 1]};
        if (CLAMP) { x.x = __builtin_amdgcn_fmed3f(x.x, -65504.f, 65504.f); x.y = __builtin_amdgcn_fmed3f(x.y, -65504.f, 65504.f); }
        const f32x2 xs = RANGE ? x * mul : x;
        const f32x2 xk = RANGE ? x * (mul * 2048.f) : x * 2048.f;
        const fp16x2_t hh = __builtin_amdgcn_cvt_pkrtz(xs.x, xs.y);
        h[2 * q] = (_Float16)hh[0]; h[2 * q + 1] = (_Float16)hh[1];
        l[2 * q] = (_Float16)fmaf((float)hh[0], -2048.f, xk.x);
        l[2 * q + 1] = (_Float16)fmaf((float)hh[1], -2048.f, xk.y);
    }
}

// acc += W-fragment (3 planes at `a`) x activation fragment (bh, bl): small terms first
__device__ __forceinline__ void mfma3(f32x16& acc, const f16x8* a, int lane, const f16x8& bh, const f16x8& bl) {
    const f16x8 ah = a[lane], al = a[64 + lane], ag = a[128 + lane];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ag, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

// ---- packed-fp32 forms of the per-element arithmetic (v_pk_mul / v_pk_add / v_pk_fma: two elements per instruction) -------------------
#define PAIR(v, q) (f32x2{(v)[2 * (q)], (v)[2 * (q) + 1]})
#define SET_PAIR(v, q, p) do { const f32x2 p_ = (p); (v)[2 * (q)] = p_.x; (v)[2 * (q) + 1] = p_.y; } while (0)
__device__ __forceinline__ f32x2 lds_pair(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// softplus(x) = max(x,0) + ln2 log2(1 + 2^(-|x| log2 e)): the arithmetic of softplus_fast (render_common.h), pairwise
__device__ __forceinline__ f32x2 softplus2(f32x2 x) {
    const f32x2 t = x * 1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(-fabsf(t.x)), __builtin_amdgcn_exp2f(-fabsf(t.y))};
    const f32x2 a = e + 1.f;
    const f32x2 lg = {__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(a.y)};
    const f32x2 mx = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return fma2(lg, f32x2{0.6931471805599453f, 0.6931471805599453f}, mx);
}
// sigmoid(x) = 1 / (1 + 2^(-x log2 e)) with the hardware reciprocal (1 ulp; the IEEE division sequence is ten instructions per value)
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {
    const f32x2 t = x * -1.4426950408889634f;
    const f32x2 a = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.f;
    return f32x2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
}
// power of two that brings `amax` to [2^target, 2^(target+1)) and its exact inverse, by exponent arithmetic (1, 1 for zero / non-finite input)
__device__ __forceinline__ void pow2_range_pair(float amax, int target_exp, float& mul, float& inv) {
    mul = 1.f; inv = 1.f;
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return;
    int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;
    e = max(-100, min(100, e));
    mul = __uint_as_float((unsigned)(127 + target_exp - e) << 23);
    inv = __uint_as_float((unsigned)(127 - target_exp + e) << 23);
}

template <bool BWD>
__device__ __forceinline__ Frags setup_frags(char* lds, const DecodeArgs& a) {
    Frags F;
    f16x8* q = reinterpret_cast<f16x8*>(lds);
    f16x8* a1 = q; f16x8* a2 = q + FRAG_PLANES; f16x8* a3 = nullptr; f16x8* a4 = nullptr;
    q += 2 * FRAG_PLANES;
    if (BWD) { a3 = q; a4 = q + FRAG_PLANES; q += 2 * FRAG_PLANES; }
    float* fl = reinterpret_cast<float*>(q);
    float* ws = fl; float* bi0 = fl + 64; float* bi1 = fl + 128; float* red = fl + 160;
    // range multipliers of the two weight matrices (every block derives the same values itself: 4 K floats)
    float m0 = 0.f, m1 = 0.f;
    for (int i = threadIdx.x; i < HD * FC; i += blockDim.x) m0 = fmaxf(m0, fabsf(a.w0[i]));
    for (int i = threadIdx.x; i < HD * (1 + CO); i += blockDim.x) m1 = fmaxf(m1, fabsf(a.w1t[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
    if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = m0; red[(threadIdx.x >> 6) * 2 + 1] = m1; }
    __syncthreads();
    m0 = m1 = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { m0 = fmaxf(m0, red[2 * w]); m1 = fmaxf(m1, red[2 * w + 1]); }          // (4 or 8 waves)
    const float s0 = pow2_range_mul(m0, 11), s1 = pow2_range_mul(m1, 11);
    F.inv0 = 1.f / s0; F.inv1 = 1.f / s1;
    // one (block/k-step, lane) fragment per thread iteration: 4 x 64 per GEMM
    for (int i = threadIdx.x; i < 4 * 64; i += blockDim.x) {
        const int lane = i & 63, bs = i >> 6;                 // bs = 2 ht + s  (layer 1 / dH)   or   ks  (layer 2 / dF)
        const int li = lane & 31, g = lane >> 5, hi = bs >> 1, s = bs & 1;
        float v1[8], v2[8], v3[8], v4[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = split_idx(8 * s + j, g);
            v1[j] = a.w0[(32 * hi + li) * FC + k] * s0;
            v2[j] = a.w1t[(32 * hi + k) * (1 + CO) + 1 + li] * s1;
            v3[j] = a.w1t[(32 * hi + li) * (1 + CO) + 1 + k] * s1;
            v4[j] = a.w0[(32 * hi + k) * FC + li] * s0;
        }
        auto put = [&](f16x8* dst, const float (&v)[8]) {
            f16x8 h, l, gg;
            split_pieces8(v, 1.f, h, l);
#pragma unroll
            for (int j = 0; j < 8; ++j) gg[j] = h[j] * (_Float16)0.00048828125f;
            dst[(bs * 3 + 0) * 64 + lane] = h; dst[(bs * 3 + 1) * 64 + lane] = l; dst[(bs * 3 + 2) * 64 + lane] = gg;
        };
        put(a1, v1); put(a2, v2);
        if (BWD) { put(a3, v3); put(a4, v4); }
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        const int h = i & 1, r = (i >> 1) & 15, ht = i >> 5;
        const int e = 32 * ht + split_idx(r, h), j = 32 * h + 16 * ht + r;
        ws[j] = a.w1t[e * (1 + CO)];
        bi0[j] = a.b0[e];
        if (ht == 0) bi1[16 * h + r] = a.b1[1 + split_idx(r, h)];
    }
    __syncthreads();
    F.a1 = a1; F.a2 = a2; F.a3 = a3; F.a4 = a4; F.ws = ws; F.bi0 = bi0; F.bi1 = bi1;
    return F;
}

// split-layout bilinear gather: this lane accumulates features split_idx(r, h), r = 0..15, of the sample at (x,y,z)
__device__ __forceinline__ void gather_split(const float* __restrict__ pn, int Hp, int Wp, int ldp, float cs, float x, float y, float z, int h,
                                             float (&f)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = 0.f;
    // one plane at a time (16 sixteen-byte loads in flight per lane): fully unrolled, the scheduler hoists all 48 loads and their
    // 192 destination registers, which pins the backward kernel at ONE wave per SIMD and leaves the gather latency fully exposed
#pragma unroll 1
    for (int pl = 0; pl < 3; ++pl) {
        float u, v;
        plane_uv(pl, x * cs, y * cs, z * cs, u, v);
        float ix = ((u + 1.f) * Wp - 1.f) * 0.5f, iy = ((v + 1.f) * Hp - 1.f) * 0.5f;
        float fx0 = floorf(ix), fy0 = floorf(iy);
        int x0 = (int)fx0, y0 = (int)fy0;
        float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
        // Branch-free: a corner outside the plane reads the clamped texel with weight 0 (zeros padding).  With the loads under a per-corner
        // branch every corner was its own basic block -- issue four loads, wait, multiply -- i.e. twelve exposed memory round trips per tile.
        float4 tv[4][4];
        float wq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
            const bool inb = (unsigned)xx < (unsigned)Wp && (unsigned)yy < (unsigned)Hp;
            const int xc = min(max(xx, 0), Wp - 1), yc = min(max(yy, 0), Hp - 1);
            const float* t = pn + (yc * Wp + xc) * ldp + pl * FC + 4 * h;
            wq[q] = inb ? ((q & 1) ? wx1 : wx0) * ((q >> 1) ? wy1 : wy0) : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) tv[q][g] = *reinterpret_cast<const float4*>(t + 8 * g);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f[4 * g + 0] = fmaf(wq[q], tv[q][g].x, f[4 * g + 0]); f[4 * g + 1] = fmaf(wq[q], tv[q][g].y, f[4 * g + 1]);
                f[4 * g + 2] = fmaf(wq[q], tv[q][g].z, f[4 * g + 2]); f[4 * g + 3] = fmaf(wq[q], tv[q][g].w, f[4 * g + 3]);
            }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = f[r] * (1.f / 3.f);      // mean over the planes (a multiplication: the IEEE division sequence is ~10 instructions per value)
}

// Compiler-level fence between the phases of a tile.  Without it the scheduler hoists the LDS fragment reads and the gather of later
// phases above the MFMA chains of earlier ones: 256 VGPRs + 92 AGPRs for the backward kernel (one wave per SIMD, every latency exposed).
// With it: 137 VGPRs, no spills.
__device__ __forceinline__ void phase_fence() { asm volatile("" ::: "memory"); }
// ... and one the instruction scheduler does not move anything across either (the GRAM steps: their operands are register-only computations
// that the scheduler otherwise starts early, all at once)
__device__ __forceinline__ void hard_fence() { asm volatile("" ::: "memory"); }

#ifndef DEC_OCC_BWD
#define DEC_OCC_BWD 3
#endif
#ifndef DEC_OCC_FWD
#define DEC_OCC_FWD 4
#endif
#ifndef DEC_GRID_BWD
#define DEC_GRID_BWD 3
#endif
#ifndef DEC_GRID_FWD
#define DEC_GRID_FWD 5
#endif
// ---- the gather as a pass of its own ---------------------------------------------------------------------------------------------------
// thread = (row, channel quad): twelve independent 16-byte texel loads, ~40 registers, every wave slot of the CU in use.  53 us for the
// 0.79 M rows of one pass (22 TB/s of texel reads out of L1 / L2) -- inside the decoder kernels the same loads cost ~95 us of a 160 us
// launch, because those hold ~140 registers (3-5 waves per SIMD) and each wave can keep only one plane's 16 loads in flight.
__global__ void __launch_bounds__(256) gather_rows_kernel(const DecodeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = i >> 3;
    const int q = (int)(i & 7);
    if (row >= a.M) return;
    const float* pp = a.pos + row * a.pos_stride;
    const float px = pp[0], py = pp[1], pz = pp[2];
    const unsigned seg = a.seg_len ? (unsigned)row / (unsigned)a.seg_len : 0u;
    const int64_t orow = a.seg_len ? (int64_t)seg * a.seg_stride + a.seg_off + ((unsigned)row - seg * (unsigned)a.seg_len) : row;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!isnan(px)) {
        const int n = (int)((unsigned)row / (unsigned)a.rows_per_image);
        const float* pn = a.planes + (int64_t)n * a.Hp * a.Wp * a.ldp + 4 * q;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            float u, v;
            plane_uv(pl, px * a.cs, py * a.cs, pz * a.cs, u, v);
            const float ix = ((u + 1.f) * a.Wp - 1.f) * 0.5f, iy = ((v + 1.f) * a.Hp - 1.f) * 0.5f;
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;      // the weights of gather_split
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
                const bool inb = (unsigned)xx < (unsigned)a.Wp && (unsigned)yy < (unsigned)a.Hp;
                const int xc = min(max(xx, 0), a.Wp - 1), yc = min(max(yy, 0), a.Hp - 1);
                const float w = inb ? ((c & 1) ? wx1 : wx0) * ((c >> 1) ? wy1 : wy0) : 0.f;
                const float4 t = *reinterpret_cast<const float4*>(pn + (yc * a.Wp + xc) * a.ldp + pl * FC);
                acc.x = fmaf(w, t.x, acc.x); acc.y = fmaf(w, t.y, acc.y); acc.z = fmaf(w, t.z, acc.z); acc.w = fmaf(w, t.w, acc.w);
            }
        }
        acc.x *= 1.f / 3.f; acc.y *= 1.f / 3.f; acc.z *= 1.f / 3.f; acc.w *= 1.f / 3.f;
    }
    float4* dst = reinterpret_cast<float4*>(a.feat + orow * FC) + q;
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    if (a.nt_rows) __builtin_nontemporal_store(f32x4n{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4n*>(dst));          // a stream larger than the 256 MB cache would only evict the planes the gather is reading
    else *dst = acc;
}

// ---- the position gradient as a pass of its own (backward, when the camera is optimised) ---------------------------------------------
// dL/d position = sum_planes sum_corners (df . texel) d weight / d position needs the twelve texels of a sample again.  Inside the
// decoder's backward kernel that re-gather costs what the forward gather did there (one plane's 16 loads in flight per wave at ~140
// registers: 214 -> 391 us for 1.57 M rows); here thread = (row, channel quad) like gather_rows_kernel: the quad of df the decoder just
// wrote, twelve independent 16-byte texel loads, three partial sums, an 8-lane reduction.
__global__ void __launch_bounds__(256) gather_grad_rows_kernel(const DecodeArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = i >> 3;
    const int q = (int)(i & 7);
    const bool in_range = row < a.M;
    float4 ps = make_float4(NAN, 0.f, 0.f, 0.f);
    if (in_range) ps = *reinterpret_cast<const float4*>(a.pos + row * 4);
    const bool valid = in_range && !isnan(ps.x);
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (valid) {
        const float4 dfq = *reinterpret_cast<const float4*>(a.df_rows + row * FC + 4 * q);
        const int n = (int)((unsigned)row / (unsigned)a.rows_per_image);
        const float* pn = a.planes + (int64_t)n * a.Hp * a.Wp * a.ldp + 4 * q;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            float u, v;
            plane_uv(pl, ps.x * a.cs, ps.y * a.cs, ps.z * a.cs, u, v);
            const float ix = ((u + 1.f) * a.Wp - 1.f) * 0.5f, iy = ((v + 1.f) * a.Hp - 1.f) * 0.5f;
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
                const bool inb = (unsigned)xx < (unsigned)a.Wp && (unsigned)yy < (unsigned)a.Hp;
                const int xc = min(max(xx, 0), a.Wp - 1), yc = min(max(yy, 0), a.Hp - 1);
                const float4 t = *reinterpret_cast<const float4*>(pn + (yc * a.Wp + xc) * a.ldp + pl * FC);
                float dot = fmaf(t.w, dfq.w, fmaf(t.z, dfq.z, fmaf(t.y, dfq.y, t.x * dfq.x)));
                dot = inb ? dot : 0.f;
                gix += dot * ((c & 1) ? 1.f : -1.f) * ((c >> 1) ? wy1 : wy0);
                giy += dot * ((c >> 1) ? 1.f : -1.f) * ((c & 1) ? wx1 : wx0);
            }
            const float gu = gix * (0.5f * a.Wp) * a.cs, gv = giy * (0.5f * a.Hp) * a.cs;
            if (pl == 0) { gx += gu; gy += gv; } else if (pl == 1) { gx += gu; gz += gv; } else { gz += gu; gx += gv; }
        }
    }
#pragma unroll
    for (int o = 1; o <= 4; o <<= 1) { gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); gz += __shfl_xor(gz, o); }
    if (q == 0 && in_range) a.gc_rows[row] = valid ? make_float4(gx, gy, gz, ps.w) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// GRAM (backward only): the two Gram products of the decoder-weight gradients, dPRE^T F [64 x 32] and dOUT^T H [32 x 64], the sigma row
// sum_s dsigma_s H[s][:] and the bias sums are accumulated here instead of dumping their four operands (1.0 GB per step) for a GEMM pass
// over them.  The contraction runs over SAMPLES, which the wave holds one per lane, so the operands are transposed through a wave-private
// 14 KB LDS area: lane = sample writes the fp16 pieces of its k-step fragments (the pieces the decoder GEMMs consume: act_frag) one half at
// a time to [unit][sample]; lane = (unit, sample octet) reads eight consecutive samples back as the A / B operand of
// v_mfma_f32_16x16x32_f16 (K = the 32 samples of the tile).  Arithmetic = the three-product form of the rest of the file; the gradient-side operand (dPRE, dOUT, dsigma) is
// brought into fp16 range by ONE power of two per 32-sample tile (samples far below the tile's largest lose nothing that the sum keeps),
// the tile's product starts from a zero accumulator and is added, scaled back, to fp32 running sums in registers: 72 MFMAs of 16 cycles
// per tile.  Round 5 contracted exact fp32 products with v_mfma_f32_32x32x2_f32: 64 MFMAs of 64 cycles per tile -- 4096 matrix-pipe
// cycles against the 1536 of the backward itself, 158 -> 383 us for the launch.  The bias sums and the sigma row are MFMAs against
// all-ones / broadcast rows (every accumulator row holds the complete sum: one register per block is kept).  One block of eight waves per
// CU (the transposition space).
constexpr int G_PT = 72;                                                  // bytes per [unit] row: 32 samples x 2 + 8 (the two sample octets of a read land on different banks)
constexpr int G_PH = 0, G_PL = 64 * G_PT, G_QH = 128 * G_PT, G_QL = 160 * G_PT, G_SH = 192 * G_PT, G_SL = 193 * G_PT;
constexpr int GRAM_WAVE_BYTES = 194 * G_PT;                               // 13968: P (64 units) and Q (32 units) as high / low pieces + the dsigma row

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// lane (sample li, half h) -> rows unit0 + split_idx(j, h), column li, of a [unit][sample] piece image; `base` = image + 4 h G_PT + 2 li.
// Written as instructions: from C++ the compiler unpacks every high half into a register of its own first (one shift and one register per
// half) instead of storing it with ds_write_b16_d16_hi.  LDS operations of a wave execute in order, so the reads that follow need no wait.
template <int OFF>
__device__ __forceinline__ void gram_put_pair(unsigned addr, int packed) {
    asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"(addr), "v"(packed), "n"(OFF), "n"(OFF + G_PT) : "memory");
}
template <int IMG, int UNIT0>
__device__ __forceinline__ void gram_put(unsigned base, const f16x8& v) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 w = __builtin_bit_cast(i32x4, v);
    gram_put_pair<IMG + (UNIT0 + 0) * G_PT>(base, w[0]);
    gram_put_pair<IMG + (UNIT0 + 2) * G_PT>(base, w[1]);
    gram_put_pair<IMG + (UNIT0 + 8) * G_PT>(base, w[2]);
    gram_put_pair<IMG + (UNIT0 + 10) * G_PT>(base, w[3]);
}
// reader lane (row rl = lane & 15, sample octet kg = lane >> 4) <- samples 8 kg .. + 7 of row `row0 + rl`: the A / B operand of
// v_mfma_f32_16x16x32_f16 (K = the 32 samples of the tile); `base` = image + rl G_PT + 16 kg
__device__ __forceinline__ f16x8 gram_get(const char* base, int row0) {
    const f16x4 lo = *reinterpret_cast<const f16x4*>(base + row0 * G_PT), hi = *reinterpret_cast<const f16x4*>(base + row0 * G_PT + 8);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// max over the wave of a non-negative value (four row steps + the four row leaders)
__device__ __forceinline__ float wave_max_nonneg(float m) {
    m = fmaxf(m, eg3d_dpp<0xB1>(m)); m = fmaxf(m, eg3d_dpp<0x4E>(m)); m = fmaxf(m, eg3d_dpp<0x141>(m)); m = fmaxf(m, eg3d_dpp<0x140>(m));
    const int b = __float_as_int(m);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
typedef float f32x4v __attribute__((ext_vector_type(4)));
// one 16 x 16 block of a tile's Gram product from a zero accumulator: Ah Bh + 2^-11 (Ah Bl + Al Bh)  (low pieces are stored x 2^11).
// 16 x 16 blocks rather than 32 x 32: four transient registers per chain instead of sixteen -- the kernel sits at the register limit of two
// waves per SIMD with its 64 running sums.
__device__ __forceinline__ f32x4v gram_tile(const f16x8& Ah, const f16x8& Al, const f16x8& Bh, const f16x8& Bl) {
    const f32x4v Z = {0.f, 0.f, 0.f, 0.f};
    f32x4v x = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, Z, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, x, 0, 0, 0);
    const f32x4v m = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, Z, 0, 0, 0);
    return x * 0.00048828125f + m;
}
// The running sums are consumed only by the NEXT tile, so the optimiser sinks their updates to the end of the loop body and keeps every tile
// product alive (in scratch: 0.5 KB per lane) across the rest of the backward.  An empty statement that reads and writes the sum pins the
// update where it is written.
__device__ __forceinline__ void pin(f32x4v& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
// column sums of a block: ones^T (Bh + 2^-11 Bl) (every accumulator row holds them)
__device__ __forceinline__ float gram_colsum(const f16x8& Bh, const f16x8& Bl) {
    const f32x4v Z = {0.f, 0.f, 0.f, 0.f};
    const _Float16 o1 = (_Float16)1.f, o11 = (_Float16)0.00048828125f;
    const f16x8 one = {o1, o1, o1, o1, o1, o1, o1, o1}, eps = {o11, o11, o11, o11, o11, o11, o11, o11};
    f32x4v m = __builtin_amdgcn_mfma_f32_16x16x32_f16(eps, Bl, Z, 0, 0, 0);
    m = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, Bh, m, 0, 0, 0);
    return m[0];
}

template <bool BWD, bool FEAT, bool GRAM = false>
__global__ void __launch_bounds__(GRAM ? 512 : 256, GRAM ? 1 : (BWD ? DEC_OCC_BWD : DEC_OCC_FWD)) decode_rows_kernel(const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const Frags F = setup_frags<BWD>(lds, a);
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    // GRAM state (dead code otherwise)
    char* const GW = lds + FRAG_BYTES_BWD + (threadIdx.x >> 6) * GRAM_WAVE_BYTES;          // this wave's transposition space
    const unsigned gput = (unsigned)(uintptr_t)(GW + 4 * h * G_PT + 2 * li);                // writer: lane = (sample li, unit half h); LDS byte address
    const char* const gget = GW + (lane & 15) * G_PT + 16 * (lane >> 4);                   // reader: lane = (row lane & 15, sample octet lane >> 4)
    // running sums: block (i, j) of d W0 = rows (hidden units) 16 i .., columns (features) 16 j ..; of d W1 = rows (colours) 16 i .., columns
    // (hidden units) 16 j ..; accumulator element r of lane (c = lane & 15, g = lane >> 4) = (row 4 g + r, column c) of its block
    f32x4v gram1[4][2], gram2[2][4];
    float run_sig[4] = {0.f, 0.f, 0.f, 0.f}, run_b0[4] = {0.f, 0.f, 0.f, 0.f}, run_b1[2] = {0.f, 0.f}, cs_ds = 0.f;      // per column c (every g holds the same sums)
    if constexpr (GRAM) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { gram1[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f}; gram2[j][i] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    }
    const int64_t ntiles = (a.M + 31) / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);

    float df_max = 0.f;
    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t row = tile * 32 + li;
        float4 ps = make_float4(NAN, 0, 0, 0);
        if (row < a.M) {
            const float* q = a.pos + row * a.pos_stride;
            ps = make_float4(q[0], q[1], q[2], a.pos_stride == 4 ? q[3] : 0.f);
        }
        const bool valid = !isnan(ps.x);
        const float px = valid ? ps.x : 0.f, py = valid ? ps.y : 0.f, pz = valid ? ps.z : 0.f;
        const int n = valid ? (int)((unsigned)row / (unsigned)a.rows_per_image) : 0;          // rows fit 31 bits (launch_decode): 32-bit divides
        const float* pn = a.planes + (int64_t)n * a.Hp * a.Wp * a.ldp;

        // output row of this sample (forward: the (pass, ray, s) -> (ray, pass, s) interleave of the save buffers)
        const unsigned seg = a.seg_len ? (unsigned)row / (unsigned)a.seg_len : 0u;
        const int64_t orow = a.seg_len ? (int64_t)seg * a.seg_stride + a.seg_off + ((unsigned)row - seg * (unsigned)a.seg_len) : row;
        f32x16 f;
        if constexpr (FEAT) {            // the gather ran as its own pass (gather_rows_kernel): this lane's 16 features are four float4 of the row
            const float* fr = a.feat + (row < a.M ? orow : 0) * FC + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(fr + 8 * g);
                f[4 * g] = v.x; f[4 * g + 1] = v.y; f[4 * g + 2] = v.z; f[4 * g + 3] = v.w;
            }
        } else {
            float fr[16];
            gather_split(pn, a.Hp, a.Wp, a.ldp, a.cs, px, py, pz, h, fr);
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = fr[r];
        }

        phase_fence();
        // ---- layer 1: PRE^T = W0 F^T + b0 ;  H = softplus(PRE) -----------------------------------------------------------
        f32x16 hid[2];
        {
            f16x8 bh[2], bl[2];
            act_frag<false, true>(f, 0, 1.f, bh[0], bl[0]);
            act_frag<false, true>(f, 1, 1.f, bh[1], bl[1]);
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hid[ht][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) mfma3(hid[ht], F.a1 + (2 * ht + s) * 192, lane, bh[s], bl[s]);
            }
        }
        {
            const f32x2 inv0 = {F.inv0, F.inv0};
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int q = 0; q < 8; ++q) SET_PAIR(hid[ht], q, softplus2(fma2(PAIR(hid[ht], q), inv0, lds_pair(F.bi0 + 32 * h + 16 * ht + 2 * q))));
        }

        phase_fence();
        // ---- layer 2: OUT^T = W1c H^T + b1c ; sigma = w1s . H + b1[0] ----------------------------------------------------
        f32x16 out;
        float sig = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = 0.f;
        static_for<0, 4>([&](auto ks) {
            f16x8 bh, bl;
            act_frag<false, false>(hid[ks.value >> 1], ks.value & 1, 1.f, bh, bl);
            if constexpr (GRAM && BWD) {          // the same pieces are the H operand of the d W1 Gram product: to [unit][sample] while they exist
                gram_put<G_PH, 16 * ks.value>(gput, bh);
                gram_put<G_PL, 16 * ks.value>(gput, bl);
            }
            mfma3(out, F.a2 + ks.value * 192, lane, bh, bl);
        });
        {
            const f32x2 inv1 = {F.inv1, F.inv1};
            f32x2 sg2 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) SET_PAIR(out, q, fma2(PAIR(out, q), inv1, lds_pair(F.bi1 + 16 * h + 2 * q)));
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int q = 0; q < 8; ++q) sg2 = fma2(lds_pair(F.ws + 32 * h + 16 * ht + 2 * q), PAIR(hid[ht], q), sg2);
            sig = sg2.x + sg2.y;
        }
        sig += __shfl_xor(sig, 32);
        sig += a.b1[0];

        if (!BWD) {
            if (valid) {
                if (h == 0) a.sigma[orow] = sig;
                float* o = a.rgb + orow * CO + 4 * h;
                const f32x2 c1 = {1.002f, 1.002f}, c0 = {-0.001f, -0.001f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2 lo = fma2(sigmoid2(PAIR(out, 2 * g)), c1, c0), hi = fma2(sigmoid2(PAIR(out, 2 * g + 1)), c1, c0);
                    *reinterpret_cast<float4*>(o + 8 * g) = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            continue;
        }

        phase_fence();
        // =========================== backward ===========================
        float2 ag = make_float2(0.f, 0.f);
        if (valid) ag = a.ag[row];
        const int64_t ray = a.ray0 + (valid ? (int64_t)((unsigned)row / (unsigned)a.samples_per_ray_row) : 0);
        const float* grgb = a.d_rgb + ray * CO + 4 * h;
        // dOUT (colours): d rgb / d out = 1.002 * s (1 - s);  dL/d rgb = 2 a d_rgb
        f32x16 dout;
        {
            const float cw = valid ? 2.f * 1.002f * ag.x : 0.f;
            const f32x2 cw2 = {cw, cw};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gv = *reinterpret_cast<const float4*>(grgb + 8 * g);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x2 sg = sigmoid2(PAIR(out, 2 * g + j));
                    const f32x2 gq = j ? f32x2{gv.z, gv.w} : f32x2{gv.x, gv.y};
                    SET_PAIR(dout, 2 * g + j, (cw2 * gq) * fma2(-sg, sg, sg));          // s (1 - s) = s - s^2
                }
            }
        }
        const float dsig = valid ? ag.y : 0.f;
        if (a.dump_dout && valid) {
            float* o = a.dump_dout + row * (1 + CO);
            if (h == 0) o[0] = dsig;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[1 + split_idx(r, h)] = dout[r];
        }
        if (a.dump_feat && valid) {
            float* o = a.dump_feat + row * FC + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 8 * g) = make_float4(f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
        }
        if (a.dump_h && valid) {          // registers 4 g .. 4 g + 3 of a split-layout vector are four consecutive elements 8 g + 4 h ..: 16-byte stores
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(a.dump_h + row * HD + 32 * ht + 8 * g + 4 * h) = make_float4(hid[ht][4 * g], hid[ht][4 * g + 1], hid[ht][4 * g + 2], hid[ht][4 * g + 3]);
        }
        if constexpr (GRAM) {
            // dOUT^T H and the sigma row dsigma^T H: P = H (written by layer 2), Q = dOUT, S = dsigma, each x its tile power of two
            float m = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(dout[r]));
            float sc, isc, ss, iss;
            pow2_range_pair(wave_max_nonneg(m), 6, sc, isc);
            pow2_range_pair(wave_max_nonneg(fabsf(dsig)), 6, ss, iss);
            if (h == 0) cs_ds += dsig;
            static_for<0, 2>([&](auto ks) {
                f16x8 qh, ql;
                act_frag<true, false>(dout, ks.value, sc, qh, ql);
                gram_put<G_QH, 16 * ks.value>(gput, qh);
                gram_put<G_QL, 16 * ks.value>(gput, ql);
                hard_fence();
            });
            {
                const float x = dsig * ss;
                const fp16x2_t hh = __builtin_amdgcn_cvt_pkrtz(x, x);
                *reinterpret_cast<_Float16*>(GW + G_SH + 2 * li) = (_Float16)hh[0];       // (lanes li and li + 32 hold the same sample)
                *reinterpret_cast<_Float16*>(GW + G_SL + 2 * li) = (_Float16)fmaf((float)hh[0], -2048.f, x * 2048.f);
            }
            hard_fence();
            f16x8 Qh[2], Ql[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Qh[i] = gram_get(gget + G_QH, 16 * i); Ql[i] = gram_get(gget + G_QL, 16 * i);
                run_b1[i] = fmaf(gram_colsum(Qh[i], Ql[i]), isc, run_b1[i]);
                pin(run_b1[i]);
            }
            const f16x8 Sh = gram_get(GW + G_SH + 16 * (lane >> 4), 0), Sl = gram_get(GW + G_SL + 16 * (lane >> 4), 0);      // every accumulator row = the dsigma row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hard_fence();
                const f16x8 Ph = gram_get(gget + G_PH, 16 * j), Pl = gram_get(gget + G_PL, 16 * j);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    gram2[i][j] += gram_tile(Qh[i], Ql[i], Ph, Pl) * isc;                 // rows = colours 16 i .., columns = hidden units 16 j ..
                    pin(gram2[i][j]);
                }
                run_sig[j] = fmaf(gram_tile(Sh, Sl, Ph, Pl)[0], iss, run_sig[j]);
                pin(run_sig[j]);
            }
        }
        phase_fence();
        // ---- dH^T = W1c^T dOUT^T + w1s dsigma ;  dPRE = dH * sigmoid(PRE) = dH * (1 - exp(-H)) ---------------------------
        // the sample's gradient column is brought to [2^6, 2^7) by its own power of two and the result column scaled back
        f32x16 dh[2];
        {
            float m = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(dout[r]));
            m = fmaxf(m, __shfl_xor(m, 32));
            float sc, isc;
            pow2_range_pair(m, 6, sc, isc);
            isc *= F.inv1;
            f16x8 bh[2], bl[2];
            act_frag<true, false>(dout, 0, sc, bh[0], bl[0]);
            act_frag<true, false>(dout, 1, sc, bh[1], bl[1]);
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[ht][r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) mfma3(dh[ht], F.a3 + (2 * ht + s) * 192, lane, bh[s], bl[s]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {          // d softplus = 1 - exp(-H)
                    const f32x2 t = PAIR(hid[ht], q) * -1.4426950408889634f;
                    const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    const f32x2 lin = fma2(PAIR(dh[ht], q), f32x2{isc, isc}, lds_pair(F.ws + 32 * h + 16 * ht + 2 * q) * dsig);
                    SET_PAIR(dh[ht], q, lin * (1.f - e));
                }
            }
        }
        if (a.dump_dpre && valid) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(a.dump_dpre + row * HD + 32 * ht + 8 * g + 4 * h) = make_float4(dh[ht][4 * g], dh[ht][4 * g + 1], dh[ht][4 * g + 2], dh[ht][4 * g + 3]);
        }
        if constexpr (GRAM) {
            // dPRE^T F: P = dPRE x (tile power of two), Q = F (the fragments layer 1 consumed)
            float m = 0.f;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(dh[ht][r]));
            float sc, isc;
            pow2_range_pair(wave_max_nonneg(m), 6, sc, isc);
            static_for<0, 4>([&](auto ks) {
                f16x8 ph, pl;
                act_frag<true, false>(dh[ks.value >> 1], ks.value & 1, sc, ph, pl);
                gram_put<G_PH, 16 * ks.value>(gput, ph);
                gram_put<G_PL, 16 * ks.value>(gput, pl);
                hard_fence();
            });
            static_for<0, 2>([&](auto ks) {
                f16x8 qh, ql;
                act_frag<false, true>(f, ks.value, 1.f, qh, ql);
                gram_put<G_QH, 16 * ks.value>(gput, qh);
                gram_put<G_QL, 16 * ks.value>(gput, ql);
            });
            hard_fence();
            f16x8 Qh[2], Ql[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { Qh[j] = gram_get(gget + G_QH, 16 * j); Ql[j] = gram_get(gget + G_QL, 16 * j); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hard_fence();
                const f16x8 Ph = gram_get(gget + G_PH, 16 * i), Pl = gram_get(gget + G_PL, 16 * i);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    gram1[i][j] += gram_tile(Ph, Pl, Qh[j], Ql[j]) * isc;                 // rows = hidden units 16 i .., columns = features 16 j ..
                    pin(gram1[i][j]);
                }
                run_b0[i] = fmaf(gram_colsum(Ph, Pl), isc, run_b0[i]);
                pin(run_b0[i]);
            }
            hard_fence();
        }
        phase_fence();
        // ---- dF^T = W0^T dPRE^T -------------------------------------------------------------------------------------------
        f32x16 df;
        {
            float m = 0.f;
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(dh[ht][r]));
            m = fmaxf(m, __shfl_xor(m, 32));
            float sc, isc;
            pow2_range_pair(m, 6, sc, isc);
            isc = F.inv0 * isc * (1.f / 3.f);                                                       // / 3: mean over the three planes
#pragma unroll
            for (int r = 0; r < 16; ++r) df[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 bh, bl;
                act_frag<true, false>(dh[ks >> 1], ks & 1, sc, bh, bl);
                mfma3(df, F.a4 + ks * 192, lane, bh, bl);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) SET_PAIR(df, q, PAIR(df, q) * isc);
        }
        if (a.df_rows && valid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) df_max = fmaxf(df_max, fabsf(df[r]));
            float* o = a.df_rows + row * FC + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 8 * g) = make_float4(df[4 * g], df[4 * g + 1], df[4 * g + 2], df[4 * g + 3]);
        }
        if (a.gc_rows) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 1
            for (int pl = 0; pl < 3; ++pl) {
                float u, v;
                plane_uv(pl, px * a.cs, py * a.cs, pz * a.cs, u, v);
                float ix = ((u + 1.f) * a.Wp - 1.f) * 0.5f, iy = ((v + 1.f) * a.Hp - 1.f) * 0.5f;
                float fx0 = floorf(ix), fy0 = floorf(iy);
                int x0 = (int)fx0, y0 = (int)fy0;
                float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
                float gix = 0.f, giy = 0.f;
                float4 tv[4][4];                   // branch-free as in gather_split: clamped texel, zero contribution outside the plane
                float mk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
                    mk[q] = ((unsigned)xx < (unsigned)a.Wp && (unsigned)yy < (unsigned)a.Hp) ? 1.f : 0.f;
                    const int xc = min(max(xx, 0), a.Wp - 1), yc = min(max(yy, 0), a.Hp - 1);
                    const float* t = pn + (yc * a.Wp + xc) * a.ldp + pl * FC + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) tv[q][g] = *reinterpret_cast<const float4*>(t + 8 * g);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dot = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        dot = fmaf(tv[q][g].x, df[4 * g], dot); dot = fmaf(tv[q][g].y, df[4 * g + 1], dot);
                        dot = fmaf(tv[q][g].z, df[4 * g + 2], dot); dot = fmaf(tv[q][g].w, df[4 * g + 3], dot);
                    }
                    dot *= mk[q];
                    const float sx = (q & 1) ? 1.f : -1.f, sy = (q >> 1) ? 1.f : -1.f;
                    gix += dot * sx * ((q >> 1) ? wy1 : wy0);
                    giy += dot * sy * ((q & 1) ? wx1 : wx0);
                }
                const float gu = gix * (0.5f * a.Wp) * a.cs, gv = giy * (0.5f * a.Hp) * a.cs;
                if (pl == 0) { gx += gu; gy += gv; } else if (pl == 1) { gx += gu; gz += gv; } else { gz += gu; gx += gv; }
            }
            gx += __shfl_xor(gx, 32); gy += __shfl_xor(gy, 32); gz += __shfl_xor(gz, 32);     // the two feature halves of the sample
            if (h == 0 && row < a.M) a.gc_rows[row] = valid ? make_float4(gx, gy, gz, ps.w) : make_float4(0, 0, 0, 0);
        }
    }
    if constexpr (GRAM) {
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    eg3d_acc(a.gram_w0 + (16 * i + 4 * g + r) * FC + 16 * j + c, gram1[i][j][r] * a.gram_s0);                 // d W0 [unit][feature]
                    eg3d_acc(a.gram_w1 + (1 + 16 * j + 4 * g + r) * HD + 16 * i + c, gram2[j][i][r] * a.gram_s1);           // d W1 [1 + colour][hidden]
                }
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                eg3d_acc(a.gram_w1 + 16 * i + c, run_sig[i] * a.gram_s1);             // sigma row of d W1
                eg3d_acc(a.gram_b0 + 16 * i + c, run_b0[i] * a.gram_sb);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) eg3d_acc(a.gram_b1 + 1 + 16 * j + c, run_b1[j] * a.gram_sb);
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) cs_ds += __shfl_xor(cs_ds, o);               // (lanes 32 .. 63 hold zeros)
        if (lane == 0) eg3d_acc(a.gram_b1, cs_ds * a.gram_sb);
    }
    if constexpr (BWD) {
        if (a.df_amax != nullptr) {       // non-negative floats order like their bit patterns
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) df_max = fmaxf(df_max, __shfl_xor(df_max, o));
            if (lane == 0 && df_max < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(a.df_amax), __float_as_uint(df_max));
        }
    }
}

int launch_decode(const DecodeArgs& a, bool bwd, hipStream_t st) {
    if (a.M <= 0) return EG3D_OK;
    if (a.M > INT32_MAX || a.rows_per_image > INT32_MAX || a.samples_per_ray_row > INT32_MAX || (int64_t)a.Hp * a.Wp * a.ldp > INT32_MAX)
        return EG3D_ERR_UNSUPPORTED;      // 32-bit row and texel-offset arithmetic in the kernels
    const int64_t ntiles = (a.M + 31) / 32;
    const int blocks = (int)std::min<int64_t>((ntiles + 3) / 4, 256 * (bwd ? DEC_GRID_BWD : DEC_GRID_FWD));     // persistent: resident blocks per CU x 256 CUs
    static const bool gc_split = [] { const char* e = getenv("EG3D_GC_SPLIT"); return e ? atoi(e) != 0 : true; }();
    if (bwd && gc_split && a.gc_rows != nullptr && a.df_rows != nullptr && a.pos_stride == 4) {
        // position gradient from the df rows this launch writes, in a high-occupancy pass of its own (gather_grad_rows_kernel)
        DecodeArgs b = a;
        b.gc_rows = nullptr;
        if (int rc = launch_decode(b, true, st)) return rc;
        hipLaunchKernelGGL(gather_grad_rows_kernel, dim3((unsigned)((a.M * 8 + 255) / 256)), dim3(256), 0, st, a);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
    if (a.feat != nullptr) {
        if (!bwd) {
            DecodeArgs g = a;
            // (round 6, A/B in one session: 8 images per GPU 312.2 -> 313.6 image-steps/s with non-temporal row stores, one image 230.8 -> 230.3: only above the cache size)
            g.nt_rows = a.M * FC * 4 > (int64_t)192 << 20;
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((a.M * 8 + 255) / 256)), dim3(256), 0, st, g);
        }
        if (bwd && a.gram_w0 != nullptr) {
            static std::atomic<uint64_t> attr_done{0};
            auto kern = decode_rows_kernel<true, true, true>;
            const int smem = FRAG_BYTES_BWD + 8 * GRAM_WAVE_BYTES;                                                         // 156 KB: one resident block of eight waves per CU
            if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, attr_done)) return e;
            hipLaunchKernelGGL(kern, dim3((int)std::min<int64_t>((ntiles + 7) / 8, 256)), dim3(512), smem, st, a);
        } else if (bwd) hipLaunchKernelGGL((decode_rows_kernel<true, true>), dim3(blocks), dim3(256), FRAG_BYTES_BWD, st, a);
        else hipLaunchKernelGGL((decode_rows_kernel<false, true>), dim3(blocks), dim3(256), FRAG_BYTES_FWD, st, a);
    } else if (bwd) hipLaunchKernelGGL((decode_rows_kernel<true, false>), dim3(blocks), dim3(256), FRAG_BYTES_BWD, st, a);
    else hipLaunchKernelGGL((decode_rows_kernel<false, false>), dim3(blocks), dim3(256), FRAG_BYTES_FWD, st, a);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

}  // namespace

// ---- internal entry points used by renderer.hip (same shared object) -----------------------------------------------------
int eg3d_decode_rows_fwd(const eg3d_render_params& p, const float* pos, int pos_stride, int64_t M, int64_t rows_per_image, float* sigma, float* rgb,
                         void* stream, int seg_len, int seg_stride, int seg_off) {
    DecodeArgs a = {};
    a.planes = p.planes; a.N = p.N; a.Hp = p.Hp; a.Wp = p.Wp; a.ldp = p.ldp; a.cs = 2.f / p.box_warp;
    a.w0 = p.w0; a.b0 = p.b0; a.w1t = p.w1; a.b1 = p.b1;
    a.pos = pos; a.pos_stride = pos_stride; a.M = M; a.rows_per_image = rows_per_image; a.sigma = sigma; a.rgb = rgb;
    a.seg_len = seg_len; a.seg_stride = seg_stride; a.seg_off = seg_off;
    a.feat = p.feat_rows;
    return launch_decode(a, false, (hipStream_t)stream);
}

int eg3d_decode_rows_bwd(const eg3d_render_bwd_params& bp, const float* pos, int64_t row0, int64_t M, int64_t rows_per_image, int64_t samples_per_ray_row,
                         void* stream) {
    const eg3d_render_params& p = bp.fwd;
    DecodeArgs a = {};
    a.planes = p.planes; a.N = p.N; a.Hp = p.Hp; a.Wp = p.Wp; a.ldp = p.ldp; a.cs = 2.f / p.box_warp;
    a.w0 = p.w0; a.b0 = p.b0; a.w1t = p.w1; a.b1 = p.b1;
    a.pos = pos + row0 * 4; a.pos_stride = 4; a.M = M; a.rows_per_image = rows_per_image;
    a.ag = reinterpret_cast<const float2*>(bp.ag_rows) + row0; a.d_rgb = bp.d_rgb; a.samples_per_ray_row = samples_per_ray_row; a.ray0 = 0;
    a.df_rows = bp.df_rows ? bp.df_rows + row0 * FC : nullptr;
    a.gc_rows = bp.gc_rows ? reinterpret_cast<float4*>(bp.gc_rows) + row0 : nullptr;
    a.dump_dpre = bp.dump_dpre ? bp.dump_dpre + row0 * HD : nullptr;
    a.dump_h = bp.dump_h ? bp.dump_h + row0 * HD : nullptr;
    a.dump_dout = bp.dump_dout ? bp.dump_dout + row0 * (1 + CO) : nullptr;
    a.dump_feat = bp.dump_feat ? bp.dump_feat + row0 * FC : nullptr;
    a.feat = p.feat_rows ? p.feat_rows + row0 * FC : nullptr;
    a.df_amax = bp.df_amax;
    if (bp.gram_w0 != nullptr) {
        if (!bp.gram_b0 || !bp.gram_w1 || !bp.gram_b1 || !p.feat_rows || bp.dump_dpre || bp.dump_h || bp.dump_dout) return EG3D_ERR_INVALID;
        a.gram_w0 = bp.gram_w0; a.gram_b0 = bp.gram_b0; a.gram_w1 = bp.gram_w1; a.gram_b1 = bp.gram_b1;
        a.gram_s0 = bp.gram_scale0; a.gram_s1 = bp.gram_scale1; a.gram_sb = bp.gram_bias_scale;
    }
    return launch_decode(a, true, (hipStream_t)stream);
}
