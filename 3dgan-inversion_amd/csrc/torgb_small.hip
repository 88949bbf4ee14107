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

__global__ void __launch_bounds__(256) torgb_small_kernel(const eg3d_torgb_small_params p) {
    __shared__ float red[4][TS_PIX][TS_OUT + 1];
    const int HW = p.H * p.W;
    const int tiles = (HW + TS_PIX - 1) / TS_PIX;
    const int n = blockIdx.x / tiles, p0 = (blockIdx.x - n * tiles) * TS_PIX, o0 = blockIdx.y * TS_OUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    const int pix = p0 + row;
    const bool pok = pix < HW;
    // this wave's share of the contraction: channels [k0, k1), a multiple of 8 wide
    const int groups = p.C / 8;
    const int g0 = (int)((int64_t)groups * wave / 4), g1 = (int)((int64_t)groups * (wave + 1) / 4);
    const float* xr = p.x + ((int64_t)n * HW + (pok ? pix : 0)) * p.ldx + 4 * h;
    const float* wr = p.w + (int64_t)(o0 + row) * p.w_row + 4 * h;           // B operand: lane = (output column, k half)
    const float* sr = p.s + (int64_t)n * p.C + 4 * h;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int g = g0; g < g1; g += TS_BATCH) {
        float4 xa[TS_BATCH], wb[TS_BATCH], sv[TS_BATCH];
#pragma unroll
        for (int j = 0; j < TS_BATCH; ++j) {
            const bool ok = g + j < g1;
            const int kb = (ok ? g + j : g0) * 8;
            xa[j] = (ok && pok) ? *reinterpret_cast<const float4*>(xr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            wb[j] = ok ? *reinterpret_cast<const float4*>(wr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            sv[j] = *reinterpret_cast<const float4*>(sr + kb);
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
    // accumulator element r of a lane: pixel row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), output column lane & 31
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][row] = acc[r];
    __syncthreads();
    // epilogue: thread = (pixel row, quad of outputs)
    const int er = threadIdx.x >> 3, eq = threadIdx.x & 7;
    const int ep = p0 + er, eo = o0 + eq * 4;
    if (ep >= HW) return;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (red[0][er][eq * 4 + q] + red[1][er][eq * 4 + q]) + (red[2][er][eq * 4 + q] + red[3][er][eq * 4 + q]);
    if (p.bias != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + eo);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.clamp >= 0.f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fminf(fmaxf(v[q], -p.clamp), p.clamp);
    }
    if (p.addend != nullptr) {
        float4 a;
        if (p.addend_up2) {
            const int yy = ep / p.W, xx = ep - yy * p.W;
            a = up2_at(p.addend + (int64_t)n * (HW >> 2) * p.ldo + eo, p.ldo, p.H >> 1, p.W >> 1, yy, xx, p.addend_taps);
        } else {
            a = *reinterpret_cast<const float4*>(p.addend + ((int64_t)n * HW + ep) * p.ldo + eo);
        }
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    }
    *reinterpret_cast<float4*>(p.out + ((int64_t)n * HW + ep) * p.ldo + eo) = make_float4(v[0], v[1], v[2], v[3]);
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int eg3d_torgb_small_supported(const eg3d_torgb_small_params* p) {
    if (!p || !p->x || !p->w || !p->s || !p->out) return 0;
    if (p->N < 1 || p->H < 1 || p->W < 1 || p->C < 32 || (p->C & 7) || p->Cp < TS_OUT || (p->Cp % TS_OUT)) return 0;
    if ((p->ldx & 3) || p->ldx < p->C || (p->ldo & 3) || p->ldo < p->Cp || (p->w_row & 3) || p->w_row < p->C) return 0;
    if (!al16(p->x) || !al16(p->w) || !al16(p->s) || !al16(p->out) || (p->bias && !al16(p->bias)) || (p->addend && !al16(p->addend))) return 0;
    if (p->addend && p->addend_up2 && ((p->H & 1) || (p->W & 1))) return 0;
    if ((int64_t)p->N * eg3d_cdiv((int64_t)p->H * p->W, TS_PIX) > 0x7fffffff) return 0;
    return 1;
}

extern "C" int eg3d_torgb_small_fwd(const eg3d_torgb_small_params* p, void* stream) {
    if (!p || !p->x || !p->w || !p->s || !p->out) return EG3D_ERR_INVALID;
    if (!eg3d_torgb_small_supported(p)) return EG3D_ERR_UNSUPPORTED;
    const dim3 grid(p->N * eg3d_cdiv((int64_t)p->H * p->W, TS_PIX), p->Cp / TS_OUT);
    hipLaunchKernelGGL(torgb_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
