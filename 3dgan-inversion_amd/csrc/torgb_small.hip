// Low-latency toRGB for the small resolutions of the backbone (4^2 .. 64^2): the modulated 1 x 1 conv + bias (+ clamp) + skip image
// (reference: ToRGBLayer.forward, training/networks_stylegan2.py:338-359, with `img = upsample2d(img); img = img.add_(y)` of
// SynthesisBlock.forward :433-436) as ONE short launch.
//
// The implicit-GEMM kernel walks the 512-channel contraction in 32 barrier-separated steps whatever the pixel count: ~21 us for 64 pixels
// x 96 outputs (0.006 GFLOP).  Here the contraction is cut four ways across the waves of a block, every wave issues ALL the loads of its
// 128-channel share before the first matrix instruction (two batches of eight 16-byte loads per operand: the launch sees two memory
// latencies, not 32), multiplies on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation -- no operand
// split), and the four partial 32 x 32 tiles meet in LDS, where the epilogue adds bias, clamps, adds the skip image (full resolution, or the
// half-resolution image through the 2 x 2 taps of the separable up-sampling FIR) and stores 16 bytes per lane.
//   block = 256 threads = 4 waves; tile = 32 pixels (of ONE image) x 32 outputs; grid = (N * ceil(HW / 32), Cp / 32).
#include "common.h"
#include "det.h"
#include <algorithm>
#include <atomic>

namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// upfirdn2d.upsample2d at one output pixel for a separable 4-tap filter t (true convolution, padding (2, 1), zero outside); see conv_igemm.hip
__device__ __forceinline__ float4 up2_at(const float* __restrict__ low, int ld, int Hl, int Wl, int y, int x, const float* t) {
    const int oy = y & 1, ox = x & 1;
    const int r0 = (y >> 1) - 1 + oy, c0 = (x >> 1) - 1 + ox;
    const float wy0 = oy ? t[2] : t[3], wy1 = oy ? t[0] : t[1], wx0 = ox ? t[2] : t[3], wx1 = ox ? t[0] : t[1];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ra = r0 >= 0, rb = r0 + 1 < Hl, ca = c0 >= 0, cb = c0 + 1 < Wl;
    const float* q = low + ((int64_t)r0 * Wl + c0) * ld;
    const float4 A = (ra && ca) ? *reinterpret_cast<const float4*>(q) : z, B = (ra && cb) ? *reinterpret_cast<const float4*>(q + ld) : z;
    const float4 Cc = (rb && ca) ? *reinterpret_cast<const float4*>(q + (int64_t)Wl * ld) : z;
    const float4 D = (rb && cb) ? *reinterpret_cast<const float4*>(q + (int64_t)Wl * ld + ld) : z;
    return make_float4(wy0 * (wx0 * A.x + wx1 * B.x) + wy1 * (wx0 * Cc.x + wx1 * D.x), wy0 * (wx0 * A.y + wx1 * B.y) + wy1 * (wx0 * Cc.y + wx1 * D.y),
                       wy0 * (wx0 * A.z + wx1 * B.z) + wy1 * (wx0 * Cc.z + wx1 * D.z), wy0 * (wx0 * A.w + wx1 * B.w) + wy1 * (wx0 * Cc.w + wx1 * D.w));
}

constexpr int TS_PIX = 32, TS_OUT = 32, TS_BATCH = 8;        // TS_BATCH: 8-channel groups loaded ahead per operand (8 x 16 bytes per lane)

// NW waves per block share the contraction: 4, or 8 when a wave's share would otherwise be more than one batch of TS_BATCH channel groups (C = 512: two
// batches = two memory round trips in sequence, 2 - 2.5 us each, in a launch whose arithmetic is 1 us)
template <int NW>
__global__ void __launch_bounds__(NW * 64) torgb_small_kernel(const eg3d_torgb_small_params p) {
    __shared__ float red[NW][TS_PIX][TS_OUT + 1];
    const int HW = p.H * p.W;
    const int tiles = (HW + TS_PIX - 1) / TS_PIX;
    const int n = blockIdx.x / tiles, p0 = (blockIdx.x - n * tiles) * TS_PIX, o0 = blockIdx.y * TS_OUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    const int pix = p0 + row;
    const bool pok = pix < HW;
    // this wave's share of the contraction: channels [k0, k1), a multiple of 8 wide
    const int groups = p.C / 8;
    const int g0 = (int)((int64_t)groups * wave / NW), g1 = (int)((int64_t)groups * (wave + 1) / NW);
    const float* xr = p.x + ((int64_t)n * HW + (pok ? pix : 0)) * p.ldx + 4 * h;
    const float* wr = p.w + (int64_t)(o0 + row) * p.w_row + 4 * h;           // B operand: lane = (output column, k half)
    const float* sr = p.s + (int64_t)n * p.C + 4 * h;
    // PRE (pre_z set): the layer input does not exist yet -- the producing 3 x 3 layer left its split-K sums in pre_z and this launch runs that
    // layer's finishing epilogue while it loads its operand, x = clamp(lrelu(z d + noise + bias) gain) (the arithmetic of epilogue_fwd_kernel, in its
    // order), and the workgroups of output tile 0 write x for the layers that follow: the 5 - 10 us finishing launch of the 4^2 .. 64^2 blocks is gone
    const bool pre = p.pre_z != nullptr;
    const float* zr = pre ? p.pre_z + ((int64_t)n * HW + (pok ? pix : 0)) * p.ldx + 4 * h : nullptr;
    // (absent d / bias: the loads still go out -- to the styles, a valid address -- and their values are replaced: a load under a branch is a
    //  memory round trip of its own, sixteen of them in sequence per batch made the merged launch 8 us slower than the two it replaces)
    const bool has_d = pre && p.pre_d != nullptr, has_b = pre && p.pre_bias != nullptr;
    const float* dr = has_d ? p.pre_d + (int64_t)n * p.C + 4 * h : sr;
    const float* br = has_b ? p.pre_bias + 4 * h : sr;
    float* xw = (pre && blockIdx.y == 0 && pok) ? const_cast<float*>(p.x) + ((int64_t)n * HW + pix) * p.ldx + 4 * h : nullptr;
    const int er = (threadIdx.x >> 3) & 31, eq = threadIdx.x & 7;                   // (the epilogue is the first 256 threads' business)
    const int ep = p0 + er, eo = o0 + eq * 4;
    const bool eok = ep < HW && threadIdx.x < 256;
    const int epc = eok ? ep : 0;
    const int e_yy = epc / p.W, e_xx = epc - e_yy * p.W;          // (the division sequence before the first load goes out: its temporaries made the compiler wait for pending loads)
    // (every side input below is only REQUESTED here -- the arithmetic on it comes after the first operand batch has been requested too: as first written
    //  -- noise times strength, the four up-sampling taps combined on the spot -- the launch waited out three memory round trips of 2 - 2.5 us each before
    //  its operand loads went out, half of a 14 us launch)
    float pre_nz_raw = 0.f, pre_strength = 0.f, pre_amax = 0.f;
    if (pre && p.pre_noise != nullptr) { pre_nz_raw = p.pre_noise[pok ? (int64_t)n * p.pre_noise_nstride + pix : 0]; pre_strength = *p.pre_strength; }
    // the epilogue's side inputs do not depend on the products: issued here, they travel with the operand loads instead of after the matrix phase
    float4 bias4 = *reinterpret_cast<const float4*>((p.bias != nullptr ? p.bias : p.s) + (p.bias != nullptr ? eo : 0));          // (a valid address either way)
    // addend: the pixel itself, or the four half-resolution neighbours of the up-sampling (clamped addresses; the zero padding is applied to the VALUES later)
    // One unconditional set of four loads whatever the mode (pointer selects, no branch: behind a branch the register allocator reused a pending load's
    // destination in the other arm and the compiler waited for everything in flight)
    const bool add_up2 = p.addend != nullptr && p.addend_up2 != 0, add_full = p.addend != nullptr && !add_up2;
    bool mA = false, mB = false, mC = false, mD = false;                       // A .. D inside the half-resolution image
    const float* qA = p.s; const float* qB = p.s; const float* qC = p.s; const float* qD = p.s;          // (a valid address when there is nothing to add)
    {
        const int Hl = p.H >> 1, Wl = p.W >> 1;
        const int oy = e_yy & 1, ox = e_xx & 1;
        const int r0 = (e_yy >> 1) - 1 + oy, c0 = (e_xx >> 1) - 1 + ox;
        const bool ra = r0 >= 0, rb = r0 + 1 < Hl, ca = c0 >= 0, cb = c0 + 1 < Wl;
        const int rr0 = ra ? r0 : 0, rr1 = rb ? r0 + 1 : max(Hl - 1, 0), cc0 = ca ? c0 : 0, cc1 = cb ? c0 + 1 : max(Wl - 1, 0);
        const float* low = (p.addend != nullptr ? p.addend : p.s) + (add_up2 ? (int64_t)n * (HW >> 2) * p.ldo + eo : 0);
        if (add_up2) {
            qA = low + ((int64_t)rr0 * Wl + cc0) * p.ldo; qB = low + ((int64_t)rr0 * Wl + cc1) * p.ldo;
            qC = low + ((int64_t)rr1 * Wl + cc0) * p.ldo; qD = low + ((int64_t)rr1 * Wl + cc1) * p.ldo;
            mA = ra && ca; mB = ra && cb; mC = rb && ca; mD = rb && cb;
        } else if (add_full) {
            qA = p.addend + ((int64_t)n * HW + epc) * p.ldo + eo;
        }
    }
    const float4 adA = *reinterpret_cast<const float4*>(qA), adB = *reinterpret_cast<const float4*>(qB), adC = *reinterpret_cast<const float4*>(qC),
                 adD = *reinterpret_cast<const float4*>(qD);
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int g = g0; g < g1; g += TS_BATCH) {
        float4 xa[TS_BATCH], wb[TS_BATCH], sv[TS_BATCH], dvv[TS_BATCH], bvv[TS_BATCH];
#pragma unroll
        for (int j = 0; j < TS_BATCH; ++j) {
            const bool ok = g + j < g1;
            const int kb = (ok ? g + j : g0) * 8;
            xa[j] = (ok && pok) ? *reinterpret_cast<const float4*>((pre ? zr : xr) + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            wb[j] = ok ? *reinterpret_cast<const float4*>(wr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            sv[j] = *reinterpret_cast<const float4*>(sr + kb);
            if (pre) { dvv[j] = *reinterpret_cast<const float4*>(dr + kb); bvv[j] = *reinterpret_cast<const float4*>(br + kb); }
        }
        if (pre) {
            const float pre_nz = pok ? pre_nz_raw * pre_strength : 0.f;
#pragma unroll
            for (int j = 0; j < TS_BATCH; ++j) {
                const bool ok = g + j < g1;
                const int kb = (ok ? g + j : g0) * 8;
                float4 v = xa[j];
                if (has_d) { const float4 dv = dvv[j]; v.x *= dv.x; v.y *= dv.y; v.z *= dv.z; v.w *= dv.w; }
                if (p.pre_noise != nullptr) { v.x += pre_nz; v.y += pre_nz; v.z += pre_nz; v.w += pre_nz; }
                if (has_b) { const float4 bv = bvv[j]; v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                v.x = eg3d_pwl_fwd(v.x, p.pre_slope) * p.pre_gain; v.y = eg3d_pwl_fwd(v.y, p.pre_slope) * p.pre_gain;
                v.z = eg3d_pwl_fwd(v.z, p.pre_slope) * p.pre_gain; v.w = eg3d_pwl_fwd(v.w, p.pre_slope) * p.pre_gain;
                if (p.pre_clamp >= 0.f) {
                    v.x = fminf(fmaxf(v.x, -p.pre_clamp), p.pre_clamp); v.y = fminf(fmaxf(v.y, -p.pre_clamp), p.pre_clamp);
                    v.z = fminf(fmaxf(v.z, -p.pre_clamp), p.pre_clamp); v.w = fminf(fmaxf(v.w, -p.pre_clamp), p.pre_clamp);
                }
                if (!(ok && pok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && xw != nullptr) {
                    *reinterpret_cast<float4*>(xw + kb) = v;
                    pre_amax = fmaxf(pre_amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
                xa[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < TS_BATCH; ++j) {
            // lane (row, h) holds channels kb + 4h .. + 3 of its pixel / output: instruction q contracts the channel pair (kb + q, kb + 4 + q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].x * sv[j].x, wb[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].y * sv[j].y, wb[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].z * sv[j].z, wb[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].w * sv[j].w, wb[j].w, acc, 0, 0, 0);
        }
    }
    if (pre && blockIdx.y == 0 && p.x_amax != nullptr) {          // max|x|: the operand range of the layers that read x.  One fire-and-forget atomic per wave:
#pragma unroll                                                    // a look-before-update (eg3d_commit_amax_block) is a memory round trip this launch cannot hide
        for (int o = 32; o >= 1; o >>= 1) pre_amax = fmaxf(pre_amax, __shfl_xor(pre_amax, o));
        if (lane == 0 && pre_amax > 0.f && pre_amax < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(p.x_amax), __float_as_uint(pre_amax));
    }
    // accumulator element r of a lane: pixel row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), output column lane & 31
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][row] = acc[r];
    __syncthreads();
    // epilogue: thread = (pixel row, quad of outputs)
    if (!eok) return;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = (red[0][er][eq * 4 + q] + red[1][er][eq * 4 + q]) + (red[2][er][eq * 4 + q] + red[3][er][eq * 4 + q]);
        if (NW == 8) v[q] += (red[NW - 4][er][eq * 4 + q] + red[NW - 3][er][eq * 4 + q]) + (red[NW - 2][er][eq * 4 + q] + red[NW - 1][er][eq * 4 + q]);
    }
    if (p.bias == nullptr) bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    v[0] += bias4.x; v[1] += bias4.y; v[2] += bias4.z; v[3] += bias4.w;
    if (p.clamp >= 0.f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fminf(fmaxf(v[q], -p.clamp), p.clamp);
    }
    float4 add4 = add_full ? adA : make_float4(0.f, 0.f, 0.f, 0.f);
    if (add_up2) {
        const int oy = e_yy & 1, ox = e_xx & 1;
        const float* t = p.addend_taps;
        const float wy0 = oy ? t[2] : t[3], wy1 = oy ? t[0] : t[1], wx0 = ox ? t[2] : t[3], wx1 = ox ? t[0] : t[1];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 A = mA ? adA : z, B = mB ? adB : z, Cc = mC ? adC : z, D = mD ? adD : z;
        add4 = make_float4(wy0 * (wx0 * A.x + wx1 * B.x) + wy1 * (wx0 * Cc.x + wx1 * D.x), wy0 * (wx0 * A.y + wx1 * B.y) + wy1 * (wx0 * Cc.y + wx1 * D.y),
                           wy0 * (wx0 * A.z + wx1 * B.z) + wy1 * (wx0 * Cc.z + wx1 * D.z), wy0 * (wx0 * A.w + wx1 * B.w) + wy1 * (wx0 * Cc.w + wx1 * D.w));          // up2_at's expression
    }
    *reinterpret_cast<float4*>(p.out + ((int64_t)n * HW + ep) * p.ldo + eo) = make_float4(v[0] + add4.x, v[1] + add4.y, v[2] + add4.z, v[3] + add4.w);
}

// ---- data gradient of the same layer: dx[n,p,c] = (sum_o dy[n,p,o] w[o,c]) s[n,c] (+ addend), ds[n,c] += sum_p (sum_o ...) x[n,p,c], and --
// with act_on -- the activation backward of the layer that produced x (EG3D_EPI_BWD_ACT of the conv kernels: dx then holds THAT layer's dz;
// dbias / dd / dnoise / dstrength accumulated).  Same tiling with the roles swapped: tile = 32 pixels x 32 INPUT channels, contraction over the
// Cp outputs (96: twelve 8-groups, three per wave).
__global__ void __launch_bounds__(256) torgb_small_bwd_kernel(const eg3d_torgb_small_bwd_params p) {
    __shared__ float red[4][TS_PIX][TS_OUT + 1];
    __shared__ float col[4][TS_PIX][TS_OUT + 1];          // per-cell terms of the column sums (ds, dbias, dd, the addend's ds): summed by 128 threads, no LDS atomics
    __shared__ float sc_lds[1];
    const int HW = p.H * p.W;
    const int tiles = (HW + TS_PIX - 1) / TS_PIX;
    const int n = blockIdx.x / tiles, p0 = (blockIdx.x - n * tiles) * TS_PIX, c0 = blockIdx.y * TS_OUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    const int pix = p0 + row;
    const bool pok = pix < HW;
    const int groups = p.Cp / 8;
    const int g0 = (int)((int64_t)groups * wave / 4), g1 = (int)((int64_t)groups * (wave + 1) / 4);
    const float* gr = p.dy + ((int64_t)n * HW + (pok ? pix : 0)) * p.ldg + 4 * h;
    const float* wr = p.wa + (int64_t)(c0 + row) * p.wa_row + 4 * h;
    if (threadIdx.x == 0) sc_lds[0] = 0.f;
    // the epilogue's side inputs (layer input, styles, pass-through gradient, the producing layer's d / bias / noise) are issued with the operands
    const int er = threadIdx.x >> 3, eq = threadIdx.x & 7;
    const int ep = p0 + er, eo = c0 + eq * 4;
    const bool ok = ep < HW;
    const bool act_on = p.act_on != 0;
    const eg3d_act_bwd& ab = p.act_bwd;
    const int64_t off = ((int64_t)n * HW + (ok ? ep : 0)) * p.ldx + eo;
    float4 xin4 = make_float4(0.f, 0.f, 0.f, 0.f), a4 = xin4, abd4 = make_float4(1.f, 1.f, 1.f, 1.f), abb4 = xin4;
    float nz = 0.f;
    const float4 s4 = *reinterpret_cast<const float4*>(p.s + (int64_t)n * p.C + eo);
    // add_scale: the pass-through gradient arrives UNFINISHED -- the split-K sums z of the consumer layer's data gradient -- and its finishing pass
    // (eg3d_dgrad_finish: dx = z * add_scale[n,c]; add_ds[n,c] += sum_p z x) happens here, on values this launch loads anyway.  (The load goes
    // out unconditionally, to the styles when absent: no load under a branch.)
    const float4 as4_raw = *reinterpret_cast<const float4*>((p.add_scale != nullptr ? p.add_scale : p.s) + (int64_t)n * p.C + eo);      // (replaced by ones at its use when absent: a select here makes the launch wait for the load)
    if (ok && p.xin != nullptr) xin4 = *reinterpret_cast<const float4*>(p.xin + off);
    if (ok && p.addend != nullptr) a4 = *reinterpret_cast<const float4*>(p.addend + off);
    if (act_on) {
        if (ab.d != nullptr) abd4 = *reinterpret_cast<const float4*>(ab.d + (int64_t)n * p.C + eo);
        if (ab.bias != nullptr) abb4 = *reinterpret_cast<const float4*>(ab.bias + eo);
        if (ok && ab.noise != nullptr) nz = ab.noise[(int64_t)n * ab.noise_nstride + ep];
    }
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int g = g0; g < g1; g += TS_BATCH) {
        float4 ga[TS_BATCH], wb[TS_BATCH];
#pragma unroll
        for (int j = 0; j < TS_BATCH; ++j) {
            const bool ok = g + j < g1;
            const int kb = (ok ? g + j : g0) * 8;
            ga[j] = (ok && pok) ? *reinterpret_cast<const float4*>(gr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            wb[j] = ok ? *reinterpret_cast<const float4*>(wr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < TS_BATCH; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j].x, wb[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j].y, wb[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j].z, wb[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j].w, wb[j].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][row] = acc[r];
    __syncthreads();
    eg3d_act_bwd_consts abc = {};
    if (act_on) abc = eg3d_act_bwd_setup(ab);
    float omax = 0.f;
    float4 t_ds = make_float4(0.f, 0.f, 0.f, 0.f), t_db = t_ds, t_dq = t_ds, t_da = t_ds;
    if (ok) {
        float4 v;
        v.x = (red[0][er][eq * 4 + 0] + red[1][er][eq * 4 + 0]) + (red[2][er][eq * 4 + 0] + red[3][er][eq * 4 + 0]);
        v.y = (red[0][er][eq * 4 + 1] + red[1][er][eq * 4 + 1]) + (red[2][er][eq * 4 + 1] + red[3][er][eq * 4 + 1]);
        v.z = (red[0][er][eq * 4 + 2] + red[1][er][eq * 4 + 2]) + (red[2][er][eq * 4 + 2] + red[3][er][eq * 4 + 2]);
        v.w = (red[0][er][eq * 4 + 3] + red[1][er][eq * 4 + 3]) + (red[2][er][eq * 4 + 3] + red[3][er][eq * 4 + 3]);
        if (p.ds != nullptr) t_ds = make_float4(v.x * xin4.x, v.y * xin4.y, v.z * xin4.z, v.w * xin4.w);
        if (p.add_ds != nullptr) t_da = make_float4(a4.x * xin4.x, a4.y * xin4.y, a4.z * xin4.z, a4.w * xin4.w);
        const float4 as4 = p.add_scale != nullptr ? as4_raw : make_float4(1.f, 1.f, 1.f, 1.f);
        a4 = make_float4(a4.x * as4.x, a4.y * as4.y, a4.z * as4.z, a4.w * as4.w);
        v = make_float4(v.x * s4.x + a4.x, v.y * s4.y + a4.y, v.z * s4.z + a4.z, v.w * s4.w + a4.w);
        if (act_on) {
            float4 accb4 = make_float4(0.f, 0.f, 0.f, 0.f), accd4 = accb4;
            float cs;
            v = eg3d_act_bwd_unit(abc, v, xin4, abd4, abb4, nz * abc.strength, accb4, accd4, cs);
            t_db = accb4; t_dq = accd4;
            if (ab.dnoise != nullptr || ab.dstrength != nullptr) {      // the 8 lanes of a pixel row hold this tile's 32 channels
                cs = eg3d_row_group_sum(cs, 8);
                if (eq == 0) {
                    if (ab.dnoise != nullptr) eg3d_acc(ab.dnoise + (int64_t)n * ab.dnoise_nstride + ep, cs * abc.strength);
                    if (ab.dstrength != nullptr && cs * nz != 0.f) EG3D_LDS_ACC(sc_lds, ab.dstrength, cs * nz);
                }
            }
        }
        omax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        *reinterpret_cast<float4*>(p.dx + off) = v;
    }
    eg3d_commit_amax_block(omax, p.out_amax);
    col[0][er][eq * 4 + 0] = t_ds.x; col[0][er][eq * 4 + 1] = t_ds.y; col[0][er][eq * 4 + 2] = t_ds.z; col[0][er][eq * 4 + 3] = t_ds.w;
    col[1][er][eq * 4 + 0] = t_db.x; col[1][er][eq * 4 + 1] = t_db.y; col[1][er][eq * 4 + 2] = t_db.z; col[1][er][eq * 4 + 3] = t_db.w;
    col[2][er][eq * 4 + 0] = t_dq.x; col[2][er][eq * 4 + 1] = t_dq.y; col[2][er][eq * 4 + 2] = t_dq.z; col[2][er][eq * 4 + 3] = t_dq.w;
    col[3][er][eq * 4 + 0] = t_da.x; col[3][er][eq * 4 + 1] = t_da.y; col[3][er][eq * 4 + 2] = t_da.z; col[3][er][eq * 4 + 3] = t_da.w;
    __syncthreads();
    if (threadIdx.x < 4 * TS_OUT) {          // thread = (which sum, column): 32 conflict-free LDS reads, then one global atomic
        const int a = threadIdx.x / TS_OUT, cc = threadIdx.x - a * TS_OUT, c = c0 + cc;
        const bool want = a == 0 ? p.ds != nullptr : (a == 3 ? p.add_ds != nullptr : (act_on && (a == 1 ? ab.dbias != nullptr : ab.dd != nullptr)));
        if (want) {
            float sum = 0.f;
#pragma unroll 8
            for (int r = 0; r < TS_PIX; ++r) sum += col[a][r][cc];
            if (a == 0) eg3d_acc(p.ds + (int64_t)n * p.C + c, sum);
            else if (a == 3) eg3d_acc(p.add_ds + (int64_t)n * p.C + c, sum);
            else if (a == 1) eg3d_acc(ab.dbias + c, sum);
            else eg3d_acc(ab.dd + (int64_t)n * p.C + c, sum / (ab.d != nullptr ? ab.d[(int64_t)n * p.C + c] : 1.f));      // dL/dd = sum dy * z,  z = (pre - bias - noise) / d
        }
    }
    if (act_on && ab.dstrength != nullptr && threadIdx.x == 0 && sc_lds[0] != 0.f) eg3d_acc(ab.dstrength, sc_lds[0]);
}


// =====================================================================================================================================
// The same layer for the 128^2 / 256^2 blocks of the backbone (96 outputs, 128 - 256 input channels, >= 8192 pixels): there the launch is a
// pass over x -- 17 / 34 MB in, 6 / 25 MB out, 1.6 GFLOP -- and what the small kernel does well (one memory round trip per launch) is beside the
// point: as 6144 independent 32 x 32 tiles it re-reads x once per output tile and 16 KB of weights per 16 KB of x, and its backward ends every
// tile with 128 atomics onto the same few hundred addresses (measured at 256^2: forward 73 us, backward 122 us; the implicit GEMM 42 / 56).
// The "mid" kernels are persistent streaming forms of the same arithmetic (exact fp32 products, v_mfma_f32_32x32x2_f32):
//   forward : the whole weight matrix sits in LDS as MFMA B fragments (C x 96 floats), a wave owns 32 pixels and ALL 96 outputs -- x is read
//             once --, its operand loads run four channel groups ahead of the matrix instructions, nothing is shared between the waves
//             (KS = 2: the contraction of a tile is cut across a wave pair that meets in LDS -- 128^2 has only 512 tiles for 1024 SIMDs);
//   backward: a wave owns (32 pixels, 32 input channels) with its 12 KB weight slice in registers, the result tile goes through a
//             wave-private LDS area to the (pixel, channel quad) layout of the small kernel's epilogue -- same expressions, 16 bytes per
//             lane --, and the column sums (ds, dbias, dd, the addend's ds) stay in registers across all tiles of the wave: one set of
//             atomics per block.
constexpr int TM_NT = 3;                                   // output tiles (Cp = 96)
typedef float tm_f4 __attribute__((ext_vector_type(4)));      // (staging arrays of HIP's float4 struct next to a compiler fence ended up in scratch memory)
constexpr int TM_DEPTH = 4;                                // channel groups in flight ahead of the matrix instructions

// compiler-level fence: keeps the load batches where they are written (without it the scheduler sinks every prefetch to its first use and
// splits it into dword loads, each followed by a full wait: no load is ever in flight under the matrix instructions)
__device__ __forceinline__ void tm_fence() { asm volatile("" ::: "memory"); }

// ADD: 0 = no addend, 1 = full-resolution addend, 2 = half-resolution addend through the 2 x 2 taps of the up-sampling FIR
template <int KS, int ADD, int FILL>
__global__ void __launch_bounds__(256, FILL == 12 ? 2 : 1) torgb_mid_kernel(const eg3d_torgb_small_params p) {
    extern __shared__ __attribute__((aligned(16))) char tm_lds[];
    float* const wrow = reinterpret_cast<float*>(tm_lds);                       // the weight matrix, [96][C + 4]
    const int groups = p.C / 8, wpitch = p.C + 4;
    float* const red = wrow + 96 * wpitch;                                      // KS = 2: [pair][48][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
    constexpr int TPB = 4 / KS;                                                 // tiles a block works on at a time
    const int sub = wave / KS, kp = wave % KS;
    const int HW = p.H * p.W, tpi = HW / TS_PIX;
    const int T = p.N * tpi;
    const int ng = groups / KS, g0 = kp * ng;
    const int stride = gridDim.x * TPB;
    const int iters = (T + stride - 1) / stride;
    const int Hl = p.H >> 1, Wl = p.W >> 1;
    const float t0 = p.addend_taps[0], t1 = p.addend_taps[1], t2 = p.addend_taps[2], t3 = p.addend_taps[3];
    // the tile of iteration `it` of this wave
    bool valid;
    int n, p0;
    const float* xr;
    const float* sr;
    auto set_tile = [&](int it) {
        const int tile = it * stride + blockIdx.x * TPB + sub;
        valid = tile < T;
        const int tl = valid ? tile : 0;
        n = tl / tpi; p0 = (tl - n * tpi) * TS_PIX;
        xr = p.x + ((int64_t)n * HW + p0 + col) * p.ldx + 4 * h + 8 * g0;
        sr = p.s + (int64_t)n * p.C + 4 * h + 8 * g0;
    };
    auto load_batch = [&](int gb, float4 (&xx)[TM_DEPTH], float4 (&ss)[TM_DEPTH]) {
#pragma unroll
        for (int j = 0; j < TM_DEPTH; ++j) {
            const int gi = min(gb + j, ng - 1);                                  // (past the end the last group is requested again: no load under a branch)
            xx[j] = *reinterpret_cast<const float4*>(xr + 8 * gi);
            ss[j] = *reinterpret_cast<const float4*>(sr + 8 * gi);
        }
    };
    // two batches of TM_DEPTH channel groups in flight: the loads of one are requested before the matrix instructions of the other
    float4 xa[TM_DEPTH], sa[TM_DEPTH], xb[TM_DEPTH], sb[TM_DEPTH];
    float bias_v[TM_NT];
    {   // Start-up: ONE memory round trip.  The bias, the first operand batch of the first tile and ALL fill loads of the thread are requested before
        // the first wait (as two fill batches, then the bias loads under their null-pointer branches, then the first operand batch, the launch spent
        // 7.7 - 8.8 us of 22 - 30 waiting four times).
        // Fill: the weight matrix is copied as it lies -- rows of C floats, 16 bytes per lane, whole rows per instruction; LDS row pitch C + 4 floats, so
        // that the B-fragment reads of the matrix instructions (lane = output row, 16 bytes at a fixed channel offset) fall on distinct banks.
        // (tools/proto/fill_probe.hip: 256 blocks pulling the same 96 KB this way take 1.2 - 1.7 us.)
#pragma unroll
        for (int t = 0; t < TM_NT; ++t) bias_v[t] = (p.bias != nullptr ? p.bias : p.w)[32 * t + col];
        set_tile(0);
        load_batch(0, xa, sa);
        const int q4 = p.C / 4;                                                    // float4s per weight row
        const int nb = 96 * q4 / 256;                                              // batches of 256 float4s: 3 C / 32 (3 .. 24)
        // (indices spelled out in the unrolled loops: with the address arithmetic in lambdas the staging arrays stayed in scratch memory)
        float* const dummy = red + 2 * 48 * 64;                                    // 256 float4s nobody reads: where the surplus slots of a batch are stored
        tm_f4 ta[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int i = min(j, nb - 1) * 256 + (int)threadIdx.x, o = i / q4, k = i - o * q4;
            ta[j] = *reinterpret_cast<const tm_f4*>(p.w + (int64_t)o * p.w_row + 4 * k);
        }
        if (FILL == 24) {
            tm_f4 tb[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int i = min(12 + j, nb - 1) * 256 + (int)threadIdx.x, o = i / q4, k = i - o * q4;
                tb[j] = *reinterpret_cast<const tm_f4*>(p.w + (int64_t)o * p.w_row + 4 * k);
            }
            tm_fence();
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int i = (12 + j) * 256 + (int)threadIdx.x, o = i / q4, k = i - o * q4;
                float* d = 12 + j < nb ? wrow + o * wpitch + 4 * k : dummy + 4 * (int)threadIdx.x;
                *reinterpret_cast<tm_f4*>(d) = tb[j];
            }
        }
        tm_fence();
        if (p.bias == nullptr) { bias_v[0] = 0.f; bias_v[1] = 0.f; bias_v[2] = 0.f; }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int i = j * 256 + (int)threadIdx.x, o = i / q4, k = i - o * q4;
            float* d = j < nb ? wrow + o * wpitch + 4 * k : dummy + 4 * (int)threadIdx.x;
            *reinterpret_cast<tm_f4*>(d) = ta[j];
        }
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (it > 0) {
            set_tile(it);
            load_batch(0, xa, sa);
        }
        tm_fence();
        f32x16_t acc[TM_NT];
#pragma unroll
        for (int t = 0; t < TM_NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const float* const wbase = wrow + col * wpitch + 8 * g0 + 4 * h;
        tm_f4 bn[TM_NT] = {*reinterpret_cast<const tm_f4*>(wbase), *reinterpret_cast<const tm_f4*>(wbase + 32 * wpitch), *reinterpret_cast<const tm_f4*>(wbase + 64 * wpitch)};
        auto compute = [&](int gb, const float4 (&xx)[TM_DEPTH], const float4 (&ss)[TM_DEPTH]) {
#pragma unroll
            for (int j = 0; j < TM_DEPTH; ++j) {
                const float4 a = make_float4(xx[j].x * ss[j].x, xx[j].y * ss[j].y, xx[j].z * ss[j].z, xx[j].w * ss[j].w);
                // (B fragments one group ahead of their use: read right before the matrix instructions, the LDS latency was exposed once per group)
                const tm_f4 b0 = bn[0], b1 = bn[1], b2 = bn[2];
                {
                    const float* wq = wbase + 8 * min(gb + j + 1, ng - 1);
                    bn[0] = *reinterpret_cast<const tm_f4*>(wq); bn[1] = *reinterpret_cast<const tm_f4*>(wq + 32 * wpitch); bn[2] = *reinterpret_cast<const tm_f4*>(wq + 64 * wpitch);
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b2.x, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b2.y, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b2.z, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b2.w, acc[2], 0, 0, 0);
            }
        };
        for (int gb = 0; gb < ng; gb += 2 * TM_DEPTH) {
            load_batch(gb + TM_DEPTH, xb, sb);
            tm_fence();
            compute(gb, xa, sa);
            tm_fence();
            load_batch(gb + 2 * TM_DEPTH, xa, sa);
            tm_fence();
            if (gb + TM_DEPTH < ng) compute(gb + TM_DEPTH, xb, sb);
            tm_fence();
        }
        if (KS == 2) {
            float* rp = red + (int64_t)sub * 48 * 64 + lane;
            if (kp == 1) {
#pragma unroll
                for (int t = 0; t < TM_NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rp[(t * 16 + r) * 64] = acc[t][r];
            }
            __syncthreads();
            if (kp == 0) {
#pragma unroll
                for (int t = 0; t < TM_NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += rp[(t * 16 + r) * 64];
            }
        }
        if (kp == 0 && valid) {
            // accumulator element r of a lane: pixel row (r & 3) + 8 (r >> 2) + 4 h, output column 32 t + col.  Four r (= four consecutive pixels) at a time:
            // all addend loads of the four are requested, then the arithmetic and the stores
            const int y0 = p0 / p.W, x0 = p0 - y0 * p.W;
            float* orow = p.out + ((int64_t)n * HW + p0) * p.ldo + col;
            const float* arow = ADD == 2 ? p.addend + (int64_t)n * (HW >> 2) * p.ldo + col : (ADD == 1 ? p.addend + ((int64_t)n * HW + p0) * p.ldo + col : nullptr);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float ad[4][TM_NT];
                if (ADD == 1) {
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                        for (int t = 0; t < TM_NT; ++t) ad[ri][t] = arow[(int64_t)(ri + 8 * rq + 4 * h) * p.ldo + 32 * t];
                } else if (ADD == 2) {
                    float tv[4][4][TM_NT], wgt[4][4];
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri) {
                        const int pr = ri + 8 * rq + 4 * h;
                        int xx = x0 + pr, yy = y0;
                        if (xx >= p.W) { xx -= p.W; ++yy; }                           // (W >= 32: at most one row boundary inside a tile)
                        const int oy = yy & 1, ox = xx & 1;
                        const int r0 = (yy >> 1) - 1 + oy, c0 = (xx >> 1) - 1 + ox;
                        const bool ra = r0 >= 0, rb = r0 + 1 < Hl, ca = c0 >= 0, cb = c0 + 1 < Wl;
                        const int rr0 = ra ? r0 : 0, rr1 = rb ? r0 + 1 : 0, cc0 = ca ? c0 : 0, cc1 = cb ? c0 + 1 : 0;      // clamped addresses, masked values
                        wgt[ri][0] = (ra && ca) ? 1.f : 0.f; wgt[ri][1] = (ra && cb) ? 1.f : 0.f; wgt[ri][2] = (rb && ca) ? 1.f : 0.f; wgt[ri][3] = (rb && cb) ? 1.f : 0.f;
                        const float* q[4] = {arow + ((int64_t)rr0 * Wl + cc0) * p.ldo, arow + ((int64_t)rr0 * Wl + cc1) * p.ldo,
                                             arow + ((int64_t)rr1 * Wl + cc0) * p.ldo, arow + ((int64_t)rr1 * Wl + cc1) * p.ldo};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int t = 0; t < TM_NT; ++t) tv[ri][c][t] = q[c][32 * t];
                    }
                    tm_fence();
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri) {
                        const int pr = ri + 8 * rq + 4 * h;
                        int xx = x0 + pr, yy = y0;
                        if (xx >= p.W) { xx -= p.W; ++yy; }
                        const int oy = yy & 1, ox = xx & 1;
                        const float wy0 = oy ? t2 : t3, wy1 = oy ? t0 : t1, wx0 = ox ? t2 : t3, wx1 = ox ? t0 : t1;
#pragma unroll
                        for (int t = 0; t < TM_NT; ++t) {
                            const float A = wgt[ri][0] != 0.f ? tv[ri][0][t] : 0.f, B = wgt[ri][1] != 0.f ? tv[ri][1][t] : 0.f;
                            const float Cc = wgt[ri][2] != 0.f ? tv[ri][2][t] : 0.f, D = wgt[ri][3] != 0.f ? tv[ri][3][t] : 0.f;
                            ad[ri][t] = wy0 * (wx0 * A + wx1 * B) + wy1 * (wx0 * Cc + wx1 * D);          // the expression of up2_at
                        }
                    }
                }
                if (ADD == 1) tm_fence();
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int r = ri + 4 * rq, pr = ri + 8 * rq + 4 * h;
#pragma unroll
                    for (int t = 0; t < TM_NT; ++t) {
                        float v = acc[t][r] + bias_v[t];
                        if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
                        orow[(int64_t)pr * p.ldo + 32 * t] = ADD ? v + ad[ri][t] : v;
                    }
                }
            }
        }
        if (KS == 2) __syncthreads();                                              // the pair's LDS area is rewritten by the next tile
    }
}

__global__ void __launch_bounds__(256) torgb_mid_bwd_kernel(const eg3d_torgb_small_bwd_params p) {
    __shared__ float red[4][TS_PIX][TS_OUT + 1];                                  // wave-private result tiles
    __shared__ float colsum[4][4][TS_OUT];                                        // [wave][which sum][column]
    __shared__ float str_lds[4];
    const int HW = p.H * p.W, tpi = HW / TS_PIX;
    const int n = blockIdx.z;                                                     // one image per grid plane: ds / dd / add_ds are per-image sums
    const int c0 = blockIdx.y * TS_OUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    const int er0 = lane >> 3, eq = lane & 7, eo = c0 + eq * 4;                     // epilogue: lane = (pixel row er0 + 8 k, channel quad)
    const bool act_on = p.act_on != 0;
    const eg3d_act_bwd& ab = p.act_bwd;
    constexpr int NG = 12;                                                         // Cp = 96: twelve 8-groups
    float4 wb[NG];
    {
        const float* wr = p.wa + (int64_t)(c0 + row) * p.wa_row + 4 * h;
#pragma unroll
        for (int g = 0; g < NG; ++g) wb[g] = *reinterpret_cast<const float4*>(wr + 8 * g);
    }
    const bool has_abb = act_on && ab.bias != nullptr;
    const float4 abb4_raw = *reinterpret_cast<const float4*>(has_abb ? ab.bias + eo : p.s);
    float4 t_ds = make_float4(0.f, 0.f, 0.f, 0.f), t_db = t_ds, t_dq = t_ds, t_da = t_ds;
    float omax = 0.f, dstr = 0.f;
    float (*my)[TS_OUT + 1] = red[wave];
    const float4 s4 = *reinterpret_cast<const float4*>(p.s + (int64_t)n * p.C + eo);
    const float4 as4_raw = *reinterpret_cast<const float4*>((p.add_scale != nullptr ? p.add_scale : p.s) + (int64_t)n * p.C + eo);
    const float4 abd4_raw = *reinterpret_cast<const float4*>(((act_on && ab.d != nullptr) ? ab.d : p.s) + (int64_t)n * p.C + eo);
    // the operand tile of the wave's first pixel tile is requested with the weight slice (one memory round trip before the first matrix instruction);
    // inside the loop the NEXT tile's operand loads go out as soon as the matrix instructions have consumed this one's registers, i.e. under the epilogue
    const int tstep = gridDim.x * 4;
    float4 ga[NG];
    {
        const int t0 = min((int)blockIdx.x * 4 + wave, tpi - 1);
        const float* gr = p.dy + ((int64_t)n * HW + t0 * TS_PIX + row) * p.ldg + 4 * h;
#pragma unroll
        for (int g = 0; g < NG; ++g) ga[g] = *reinterpret_cast<const float4*>(gr + 8 * g);
    }
    tm_fence();
    eg3d_act_bwd_consts abc = {};
    if (act_on) abc = eg3d_act_bwd_setup(ab);
    const float4 ones4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 abb4 = has_abb ? abb4_raw : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 as4 = p.add_scale != nullptr ? as4_raw : ones4, abd4 = (act_on && ab.d != nullptr) ? abd4_raw : ones4;
    for (int tile = blockIdx.x * 4 + wave; tile < tpi; tile += tstep) {
        const int p0 = tile * TS_PIX;
        // the epilogue's side inputs are requested before the matrix instructions
        const int64_t off0 = ((int64_t)n * HW + p0 + er0) * p.ldx + eo;
        float4 xin4[4], a4[4];
        float nz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t off = off0 + (int64_t)8 * k * p.ldx;
            xin4[k] = p.xin != nullptr ? *reinterpret_cast<const float4*>(p.xin + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            a4[k] = p.addend != nullptr ? *reinterpret_cast<const float4*>(p.addend + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            nz[k] = (act_on && ab.noise != nullptr) ? ab.noise[(int64_t)n * ab.noise_nstride + p0 + er0 + 8 * k] : 0.f;
        }
        tm_fence();
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[g].x, wb[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[g].y, wb[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[g].z, wb[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[g].w, wb[g].w, acc, 0, 0, 0);
        }
        tm_fence();
        {
            const int tn = min(tile + tstep, tpi - 1);                               // (past the end: a tile that exists, requested again -- no load under a branch)
            const float* gr = p.dy + ((int64_t)n * HW + tn * TS_PIX + row) * p.ldg + 4 * h;
#pragma unroll
            for (int g = 0; g < NG; ++g) ga[g] = *reinterpret_cast<const float4*>(gr + 8 * g);
        }
        tm_fence();
#pragma unroll
        for (int r = 0; r < 16; ++r) my[(r & 3) + 8 * (r >> 2) + 4 * h][row] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                       // wave-private area: LDS operations of one wave complete in order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int er = er0 + 8 * k;
            float4 v = make_float4(my[er][eq * 4 + 0], my[er][eq * 4 + 1], my[er][eq * 4 + 2], my[er][eq * 4 + 3]);
            const float4 xi = xin4[k];
            float4 a = a4[k];
            if (p.ds != nullptr) { t_ds.x += v.x * xi.x; t_ds.y += v.y * xi.y; t_ds.z += v.z * xi.z; t_ds.w += v.w * xi.w; }
            if (p.add_ds != nullptr) { t_da.x += a.x * xi.x; t_da.y += a.y * xi.y; t_da.z += a.z * xi.z; t_da.w += a.w * xi.w; }
            a = make_float4(a.x * as4.x, a.y * as4.y, a.z * as4.z, a.w * as4.w);
            v = make_float4(v.x * s4.x + a.x, v.y * s4.y + a.y, v.z * s4.z + a.z, v.w * s4.w + a.w);
            if (act_on) {
                float cs;
                v = eg3d_act_bwd_unit(abc, v, xi, abd4, abb4, nz[k] * abc.strength, t_db, t_dq, cs);
                if (ab.dnoise != nullptr || ab.dstrength != nullptr) {
                    cs = eg3d_row_group_sum(cs, 8);
                    if (eq == 0) {
                        if (ab.dnoise != nullptr) eg3d_acc(ab.dnoise + (int64_t)n * ab.dnoise_nstride + p0 + er, cs * abc.strength);
                        dstr += cs * nz[k];
                    }
                }
            }
            omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            *reinterpret_cast<float4*>(p.dx + off0 + (int64_t)8 * k * p.ldx) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    eg3d_commit_amax_block(omax, p.out_amax);
    // column sums: lanes that share a channel quad differ in bits 3 .. 5
    float cs16[16] = {t_ds.x, t_ds.y, t_ds.z, t_ds.w, t_db.x, t_db.y, t_db.z, t_db.w, t_dq.x, t_dq.y, t_dq.z, t_dq.w, t_da.x, t_da.y, t_da.z, t_da.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = cs16[i];
        v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        cs16[i] = v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dstr += __shfl_xor(dstr, o);
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) colsum[wave][i >> 2][lane * 4 + (i & 3)] = cs16[i];
    }
    if (lane == 0) str_lds[wave] = dstr;
    __syncthreads();
    if (threadIdx.x < 4 * TS_OUT) {
        const int a = threadIdx.x / TS_OUT, cc = threadIdx.x - a * TS_OUT, c = c0 + cc;
        const bool want = a == 0 ? p.ds != nullptr : (a == 3 ? p.add_ds != nullptr : (act_on && (a == 1 ? ab.dbias != nullptr : ab.dd != nullptr)));
        if (want) {
            const float sum = (colsum[0][a][cc] + colsum[1][a][cc]) + (colsum[2][a][cc] + colsum[3][a][cc]);
            if (a == 0) eg3d_acc(p.ds + (int64_t)n * p.C + c, sum);
            else if (a == 3) eg3d_acc(p.add_ds + (int64_t)n * p.C + c, sum);
            else if (a == 1) eg3d_acc(ab.dbias + c, sum);
            else eg3d_acc(ab.dd + (int64_t)n * p.C + c, sum / (ab.d != nullptr ? ab.d[(int64_t)n * p.C + c] : 1.f));
        }
    }
    if (act_on && ab.dstrength != nullptr && threadIdx.x == 0) {
        const float sv = (str_lds[0] + str_lds[1]) + (str_lds[2] + str_lds[3]);
        if (sv != 0.f) eg3d_acc(ab.dstrength, sv);
    }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int eg3d_torgb_small_supported(const eg3d_torgb_small_params* p) {
    if (!p || !p->x || !p->w || !p->s || !p->out) return 0;
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->C < 32 || (p->C & 7) || p->Cp < TS_OUT || (p->Cp % TS_OUT)) return 0;
    if ((p->ldx & 3) || p->ldx < p->C || (p->ldo & 3) || p->ldo < p->Cp || (p->w_row & 3) || p->w_row < p->C) return 0;
    if (!al16(p->x) || !al16(p->w) || !al16(p->s) || !al16(p->out) || (p->bias && !al16(p->bias)) || (p->addend && !al16(p->addend))) return 0;
    if (p->pre_z && (!al16(p->pre_z) || (p->pre_d && !al16(p->pre_d)) || (p->pre_bias && !al16(p->pre_bias)) || (p->pre_noise && !p->pre_strength) || !(p->pre_gain > 0.f))) return 0;
    if (p->addend && p->addend_up2 && ((p->H & 1) || (p->W & 1))) return 0;
    if ((int64_t)p->N * eg3d_cdiv((int64_t)p->H * p->W, TS_PIX) > 0x7fffffff) return 0;
    return 1;
}

// the streaming form (torgb_mid_kernel) takes the launch when the geometry is the backbone's 128^2 / 256^2 toRGB: 96 outputs, whole 32-pixel tiles
// inside an image, the weight matrix fits LDS, no pending epilogue, and there are enough tiles to fill the chip
extern "C" int eg3d_torgb_mid_supported(const eg3d_torgb_small_params* p) {
    if (!eg3d_torgb_small_supported(p) || p->pre_z != nullptr) return 0;
    if (p->Cp != 32 * TM_NT || (p->C % 32) || p->C > 256 || p->W < TS_PIX || ((int64_t)p->H * p->W) % TS_PIX) return 0;
    if ((int64_t)p->N * p->H * p->W < 8192) return 0;
    return 1;
}

static int launch_torgb_mid(const eg3d_torgb_small_params* p, hipStream_t st) {
    const int T = (int)((int64_t)p->N * p->H * p->W / TS_PIX);
    const int wbytes = 96 * (p->C + 4) * 4;
    // K split across wave pairs while whole tiles do not give every SIMD a wave (needs C / 16 channel groups a multiple of TM_DEPTH)
    const bool ks2 = T < 1024 && (p->C % 64) == 0;
    const int smem = wbytes + 2 * 48 * 64 * 4 + 4096;          // weights, the wave pairs' exchange area (KS = 2), the fill's dummy slots
    const int per_cu = std::max(1, (160 * 1024) / (smem + 1024));
    const int tpb = ks2 ? 2 : 4;
    const int blocks = std::min((T + tpb - 1) / tpb, 256 * std::min(per_cu, 2));
    static std::atomic<uint64_t> done[12];
    const int add = p->addend == nullptr ? 0 : (p->addend_up2 ? 2 : 1);
    void (*kerns[12])(const eg3d_torgb_small_params) = {
        torgb_mid_kernel<1, 0, 12>, torgb_mid_kernel<1, 1, 12>, torgb_mid_kernel<1, 2, 12>, torgb_mid_kernel<2, 0, 12>, torgb_mid_kernel<2, 1, 12>, torgb_mid_kernel<2, 2, 12>,
        torgb_mid_kernel<1, 0, 24>, torgb_mid_kernel<1, 1, 24>, torgb_mid_kernel<1, 2, 24>, torgb_mid_kernel<2, 0, 24>, torgb_mid_kernel<2, 1, 24>, torgb_mid_kernel<2, 2, 24>};
    const int ki = (p->C > 128 ? 6 : 0) + (ks2 ? 3 : 0) + add;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kerns[ki]), 160 * 1024, done[ki])) return e;
    hipLaunchKernelGGL(kerns[ki], dim3(blocks), dim3(256), smem, st, *p);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_torgb_small_fwd(const eg3d_torgb_small_params* p, void* stream) {
    if (!p || !p->x || !p->w || !p->s || !p->out) return EG3D_ERR_INVALID;
    if (!eg3d_torgb_small_supported(p)) return EG3D_ERR_UNSUPPORTED;
    if (eg3d_torgb_mid_supported(p)) return launch_torgb_mid(p, (hipStream_t)stream);
    const dim3 grid(p->N * eg3d_cdiv((int64_t)p->H * p->W, TS_PIX), p->Cp / TS_OUT);
    // (64^2 x 512 = 384 blocks: measured slower with eight waves, 14.9 -> 16.4 us)
    if (p->C / 8 > 4 * TS_BATCH && (int64_t)grid.x * grid.y <= 128) hipLaunchKernelGGL(torgb_small_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(torgb_small_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, *p);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_torgb_small_bwd_supported(const eg3d_torgb_small_bwd_params* p) {
    if (!p || !p->dy || !p->wa || !p->s || !p->dx) return 0;
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->Cp < 8 || (p->Cp & 7) || p->C < TS_OUT || (p->C % TS_OUT)) return 0;
    if ((p->ldg & 3) || p->ldg < p->Cp || (p->ldx & 3) || p->ldx < p->C || (p->wa_row & 3) || p->wa_row < p->Cp) return 0;
    if (!al16(p->dy) || !al16(p->wa) || !al16(p->s) || !al16(p->dx) || (p->xin && !al16(p->xin)) || (p->addend && !al16(p->addend))) return 0;
    if ((p->ds || p->act_on || p->add_ds) && !p->xin) return 0;
    if ((p->add_scale || p->add_ds) && (!p->addend || !p->add_scale || !al16(p->add_scale))) return 0;
    if (p->act_on) {
        const eg3d_act_bwd& ab = p->act_bwd;
        if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return 0;          // invertible piecewise-linear activations only
        if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return 0;
        if ((ab.d && !al16(ab.d)) || (ab.bias && !al16(ab.bias))) return 0;
    }
    if ((int64_t)p->N * eg3d_cdiv((int64_t)p->H * p->W, TS_PIX) > 0x7fffffff) return 0;
    return 1;
}

extern "C" int eg3d_torgb_mid_bwd_supported(const eg3d_torgb_small_bwd_params* p) {
    if (!eg3d_torgb_small_bwd_supported(p)) return 0;
    if (p->no_mid || p->Cp != 96 || ((int64_t)p->H * p->W) % TS_PIX || p->N > 65535) return 0;
    if ((int64_t)p->N * p->H * p->W < 4096) return 0;          // (the backbone's 64^2 block at one image included: 4096 pixels)
    return 1;
}

extern "C" int eg3d_torgb_small_bwd(const eg3d_torgb_small_bwd_params* p, void* stream) {
    if (!p || !p->dy || !p->wa || !p->s || !p->dx) return EG3D_ERR_INVALID;
    if (!eg3d_torgb_small_bwd_supported(p)) return EG3D_ERR_UNSUPPORTED;
    const bool mid = eg3d_torgb_mid_bwd_supported(p) != 0;
    const dim3 grid(p->N * eg3d_cdiv((int64_t)p->H * p->W, TS_PIX), p->C / TS_OUT);
    // streaming form: ~two resident blocks per CU in all (a wave walks several tiles with its column sums in registers: one set of atomics per block)
    const int tpi = (int)((int64_t)p->H * p->W / TS_PIX);
    const int planes = (p->C / TS_OUT) * p->N;
    const dim3 grid_mid(std::max(1, std::min((tpi + 3) / 4, 512 / std::max(1, planes))), p->C / TS_OUT, p->N);
    EG3D_DET_SCOPE(det, stream);
    EG3D_DET_BIND(det, p->ds, (int64_t)p->N * p->C);
    EG3D_DET_BIND(det, p->add_ds, (int64_t)p->N * p->C);
    if (p->act_on) { EG3D_DET_BIND_ACT(det, p->act_bwd, p->N, p->C, (int64_t)p->H * p->W); }
    EG3D_DET_COMMIT(det);
    if (mid) hipLaunchKernelGGL(torgb_mid_bwd_kernel, grid_mid, dim3(256), 0, (hipStream_t)stream, *p);
    else hipLaunchKernelGGL(torgb_small_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
