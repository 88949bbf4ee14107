// Sample-level kernels of the volume renderer: tri-plane gather + OSG decoder (32 -> 64 softplus -> 1+32), forward and
// backward, with the decoder on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces sample_from_planes/F.grid_sample (training/volumetric_rendering/renderer.py:55-66) + OSGDecoder.forward
// (training/triplane.py:124-136) and their autograd backward, for arbitrary sample positions ("rows").
//
// Mapping.  A wave processes tiles of 32 samples; the two lanes l and l+32 share sample l of the tile and each holds half of
// every 32-vector (split_idx, render_common.h).  All four GEMMs are computed transposed (result rows = output units, columns =
// samples), so the MFMA result layout of one layer *is* the B-operand layout of the next one:
//     PRE^T[64 x 32s] = W0 F^T          H = softplus(PRE)          OUT^T[32 x 32s] = W1c H^T       sigma = w1s . H  (VALU + 1 shuffle)
//     dH^T = W1c^T dOUT^T + w1s dsigma  dPRE = dH (1 - exp(-H))    dF^T [32 x 32s] = W0^T dPRE^T
// The weight-side (A) operands are per-lane constants; they are staged once per block in LDS in fragment order and fetched with one
// conflict-free ds_read_b32 per MFMA (a 64-cycle instruction), so they cost no registers.  128 MFMAs per 32 samples for the
// backward kernel, 64 for the forward one.  The gather is done in the same split layout: the two lanes of a sample read the two
// 64-byte halves of each 128-byte texel.
#include "render_common.h"

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
};

// LDS fragment images of the weights.  frag(ht, r)[lane] is the A operand of the k-step whose B operand is register r of the
// ht-th 16-register group.
struct Frags {
    float* a1;    // [2][16][64]  W0[32ht + (l&31)][idx(r,h)]                       (layer 1:  i = hidden, k = feature)
    float* a2;    // [2][16][64]  W1[1 + (l&31)][32ht + idx(r,h)]                   (layer 2:  i = colour, k = hidden)
    float* a3;    // [2][16][64]  W1[1 + idx(r,h)][32ht + (l&31)]                   (dH:       i = hidden, k = colour)
    float* a4;    // [2][16][64]  W0[32ht + idx(r,h)][(l&31)]                       (dF:       i = feature, k = hidden)
    float* ws;    // [2][16][2]   W1[0][32ht + idx(r,h)]                            (sigma row)
    float* bi0;   // [2][16][2]   b0[32ht + idx(r,h)]
    float* bi1;   // [16][2]      b1[1 + idx(r,h)]
};
constexpr int FRAG_FLOATS_FWD = 2 * 2048 + 64 + 64 + 32;
constexpr int FRAG_FLOATS_BWD = 4 * 2048 + 64 + 64 + 32;

template <bool BWD>
__device__ __forceinline__ Frags setup_frags(float* lds, const DecodeArgs& a) {
    Frags F;
    F.a1 = lds; F.a2 = lds + 2048;
    float* p = lds + 4096;
    if (BWD) { F.a3 = p; F.a4 = p + 2048; p += 4096; } else { F.a3 = F.a4 = nullptr; }
    F.ws = p; F.bi0 = p + 64; F.bi1 = p + 128;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) {
        const int lane = i & 63, r = (i >> 6) & 15, ht = i >> 10;
        const int li = lane & 31, e = split_idx(r, lane >> 5);
        F.a1[i] = a.w0[(32 * ht + li) * FC + e];
        F.a2[i] = a.w1t[(32 * ht + e) * (1 + CO) + 1 + li];
        if (BWD) {
            F.a3[i] = a.w1t[(32 * ht + li) * (1 + CO) + 1 + e];
            F.a4[i] = a.w0[(32 * ht + e) * FC + li];
        }
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        const int h = i & 1, r = (i >> 1) & 15, ht = i >> 5;
        const int e = 32 * ht + split_idx(r, h);
        F.ws[i] = a.w1t[e * (1 + CO)];
        F.bi0[i] = a.b0[e];
        if (ht == 0) F.bi1[i] = a.b1[1 + split_idx(r, h)];
    }
    __syncthreads();
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
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int xx = x0 + (q & 1), yy = y0 + (q >> 1);
            if ((unsigned)xx < (unsigned)Wp && (unsigned)yy < (unsigned)Hp) {
                const float* t = pn + ((int64_t)yy * Wp + xx) * ldp + pl * FC + 4 * h;
                const float w = ((q & 1) ? wx1 : wx0) * ((q >> 1) ? wy1 : wy0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 tv = *reinterpret_cast<const float4*>(t + 8 * g);
                    f[4 * g + 0] = fmaf(w, tv.x, f[4 * g + 0]); f[4 * g + 1] = fmaf(w, tv.y, f[4 * g + 1]);
                    f[4 * g + 2] = fmaf(w, tv.z, f[4 * g + 2]); f[4 * g + 3] = fmaf(w, tv.w, f[4 * g + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = f[r] / 3.f;
}

// Compiler-level fence between the phases of a tile.  Without it the scheduler hoists the LDS fragment reads and the gather of later
// phases above the MFMA chains of earlier ones: 256 VGPRs + 92 AGPRs for the backward kernel (one wave per SIMD, every latency exposed).
// With it: 137 VGPRs, no spills.
__device__ __forceinline__ void phase_fence() { asm volatile("" ::: "memory"); }

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
#define DEC_GRID_FWD 6
#endif
template <bool BWD>
__global__ void __launch_bounds__(256, BWD ? DEC_OCC_BWD : DEC_OCC_FWD) decode_rows_kernel(const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Frags F = setup_frags<BWD>(lds, a);
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int64_t ntiles = (a.M + 31) / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);

    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t row = tile * 32 + li;
        float4 ps = make_float4(NAN, 0, 0, 0);
        if (row < a.M) {
            const float* q = a.pos + row * a.pos_stride;
            ps = make_float4(q[0], q[1], q[2], a.pos_stride == 4 ? q[3] : 0.f);
        }
        const bool valid = !isnan(ps.x);
        const float px = valid ? ps.x : 0.f, py = valid ? ps.y : 0.f, pz = valid ? ps.z : 0.f;
        const int n = valid ? (int)(row / a.rows_per_image) : 0;
        const float* pn = a.planes + (int64_t)n * a.Hp * a.Wp * a.ldp;

        float f[16];
        gather_split(pn, a.Hp, a.Wp, a.ldp, a.cs, px, py, pz, h, f);

        phase_fence();
        // ---- layer 1: PRE^T = W0 F^T + b0 ;  H = softplus(PRE) -----------------------------------------------------------
        f32x16 hid[2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[ht][r] = F.bi0[(ht * 16 + r) * 2 + h];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            hid[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a1[(0 * 16 + r) * 64 + lane], f[r], hid[0], 0, 0, 0);
            hid[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a1[(1 * 16 + r) * 64 + lane], f[r], hid[1], 0, 0, 0);
        }
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[ht][r] = softplus_fast(hid[ht][r]);

        phase_fence();
        // ---- layer 2: OUT^T = W1c H^T + b1c ; sigma = w1s . H + b1[0] ----------------------------------------------------
        f32x16 out;
        float sig = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = F.bi1[r * 2 + h];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a2[(ht * 16 + r) * 64 + lane], hid[ht][r], out, 0, 0, 0);
                sig = fmaf(F.ws[(ht * 16 + r) * 2 + h], hid[ht][r], sig);
            }
        sig += __shfl_xor(sig, 32);
        sig += a.b1[0];

        if (!BWD) {
            if (valid) {
                const int64_t orow = a.seg_len ? (row / a.seg_len) * a.seg_stride + a.seg_off + row % a.seg_len : row;
                if (h == 0) a.sigma[orow] = sig;
                float* o = a.rgb + orow * CO + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(o + 8 * g) = make_float4(sigmoid_fast(out[4 * g]) * 1.002f - 0.001f, sigmoid_fast(out[4 * g + 1]) * 1.002f - 0.001f,
                                                                       sigmoid_fast(out[4 * g + 2]) * 1.002f - 0.001f, sigmoid_fast(out[4 * g + 3]) * 1.002f - 0.001f);
            }
            continue;
        }

        phase_fence();
        // =========================== backward ===========================
        float2 ag = make_float2(0.f, 0.f);
        if (valid) ag = a.ag[row];
        const int64_t ray = a.ray0 + (valid ? row / a.samples_per_ray_row : 0);
        const float* grgb = a.d_rgb + ray * CO + 4 * h;
        // dOUT (colours): d rgb / d out = 1.002 * s (1 - s);  dL/d rgb = 2 a d_rgb
        f32x16 dout;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 gv = *reinterpret_cast<const float4*>(grgb + 8 * g);
            const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sg = sigmoid_fast(out[4 * g + j]);
                dout[4 * g + j] = valid ? (2.f * ag.x * gq[j]) * 1.002f * sg * (1.f - sg) : 0.f;
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
            float* o = a.dump_feat + row * FC;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[split_idx(r, h)] = f[r];
        }
        if (a.dump_h && valid) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.dump_h[row * HD + 32 * ht + split_idx(r, h)] = hid[ht][r];
        }
        phase_fence();
        // ---- dH^T = W1c^T dOUT^T + w1s dsigma ;  dPRE = dH * sigmoid(PRE) = dH * (1 - exp(-H)) ---------------------------
        f32x16 dh[2];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[ht][r] = F.ws[(ht * 16 + r) * 2 + h] * dsig;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dh[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a3[(0 * 16 + r) * 64 + lane], dout[r], dh[0], 0, 0, 0);
            dh[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a3[(1 * 16 + r) * 64 + lane], dout[r], dh[1], 0, 0, 0);
        }
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[ht][r] = dh[ht][r] * (1.f - __expf(-hid[ht][r]));
        if (a.dump_dpre && valid) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int r = 0; r < 16; ++r) a.dump_dpre[row * HD + 32 * ht + split_idx(r, h)] = dh[ht][r];
        }
        phase_fence();
        // ---- dF^T = W0^T dPRE^T -------------------------------------------------------------------------------------------
        f32x16 df;
#pragma unroll
        for (int r = 0; r < 16; ++r) df[r] = 0.f;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 16; ++r) df = __builtin_amdgcn_mfma_f32_32x32x2f32(F.a4[(ht * 16 + r) * 64 + lane], dh[ht][r], df, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) df[r] = df[r] / 3.f;                 // mean over the three planes
        if (a.df_rows && valid) {
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
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int xx = x0 + (q & 1), yy = y0 + (q >> 1);
                    if ((unsigned)xx < (unsigned)a.Wp && (unsigned)yy < (unsigned)a.Hp) {
                        const float* t = pn + ((int64_t)yy * a.Wp + xx) * a.ldp + pl * FC + 4 * h;
                        float dot = 0.f;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 tv = *reinterpret_cast<const float4*>(t + 8 * g);
                            dot = fmaf(tv.x, df[4 * g], dot); dot = fmaf(tv.y, df[4 * g + 1], dot);
                            dot = fmaf(tv.z, df[4 * g + 2], dot); dot = fmaf(tv.w, df[4 * g + 3], dot);
                        }
                        const float sx = (q & 1) ? 1.f : -1.f, sy = (q >> 1) ? 1.f : -1.f;
                        gix += dot * sx * ((q >> 1) ? wy1 : wy0);
                        giy += dot * sy * ((q & 1) ? wx1 : wx0);
                    }
                }
                const float gu = gix * (0.5f * a.Wp) * a.cs, gv = giy * (0.5f * a.Hp) * a.cs;
                if (pl == 0) { gx += gu; gy += gv; } else if (pl == 1) { gx += gu; gz += gv; } else { gz += gu; gx += gv; }
            }
            gx += __shfl_xor(gx, 32); gy += __shfl_xor(gy, 32); gz += __shfl_xor(gz, 32);     // the two feature halves of the sample
            if (h == 0 && row < a.M) a.gc_rows[row] = valid ? make_float4(gx, gy, gz, ps.w) : make_float4(0, 0, 0, 0);
        }
    }
}

int launch_decode(const DecodeArgs& a, bool bwd, hipStream_t st) {
    if (a.M <= 0) return EG3D_OK;
    const int64_t ntiles = (a.M + 31) / 32;
    const int blocks = (int)std::min<int64_t>((ntiles + 3) / 4, 256 * (bwd ? DEC_GRID_BWD : DEC_GRID_FWD));     // persistent: resident blocks per CU x 256 CUs
    if (bwd) hipLaunchKernelGGL(decode_rows_kernel<true>, dim3(blocks), dim3(256), FRAG_FLOATS_BWD * sizeof(float), st, a);
    else hipLaunchKernelGGL(decode_rows_kernel<false>, dim3(blocks), dim3(256), FRAG_FLOATS_FWD * sizeof(float), st, a);
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
    return launch_decode(a, true, (hipStream_t)stream);
}
