// Low-latency split convolution for the smallest layers of the backbone (4^2 .. 32^2 x 512 channels): the EG3D_EPI_ATOMIC launch of
// eg3d_conv2d_igemm_f32 (partial sums added into a pre-zeroed output, finished by the epilogue passes) for few output cells.
//
// There the implicit GEMM is all latency: 64 .. 1024 cells, a 4608-deep contraction walked in barrier-separated 16-channel steps, 10 - 30 us per
// launch for 0.07 - 1.2 GFLOP.  Same recipe as csrc/torgb_small.hip: one workgroup per (32 cells, 32 output channels, TAP) -- the tap loop
// becomes grid parallelism --, the tap's channels cut four ways across the waves, every wave issues all loads of its share up front (two
// batches of eight 16-byte loads per operand at 512 channels), multiplies on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32
// products), the four partial tiles meet in LDS and go out as coalesced fp32 atomics.  Tap classes (eg3d_conv_class) are honoured in full --
// strided reads (the stride-2 adjoint of the up layers), strided writes (their transposed form), zero padding --, so the 3x3 layers, the
// up-sampling layers and all their data gradients use it.
#include "common.h"
#include "det.h"

namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
constexpr int CS_CELLS = 32, CS_OUT = 32, CS_BATCH = 8;

__global__ void __launch_bounds__(256) conv_small_kernel(const eg3d_conv_params p, const int max_tiles) {
    __shared__ float red[4][CS_CELLS][CS_OUT + 1];
    const eg3d_conv_class& cl = p.cls[blockIdx.z / 9];
    const int t = blockIdx.z % 9;
    if (t >= cl.ntaps) return;
    const int cells = cl.Ha * cl.Wa;
    const int n = blockIdx.x / max_tiles, tile = blockIdx.x - n * max_tiles;
    if (tile * CS_CELLS >= cells) return;
    const int o0 = blockIdx.y * CS_OUT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 31, h = lane >> 5;
    const int m = tile * CS_CELLS + row;
    const int ay = m / cl.Wa, ax = m - ay * cl.Wa;
    const int iy = ay * p.in_stride + cl.dy[t], ix = ax * p.in_stride + cl.dx[t];
    const bool pok = m < cells && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    const int groups = p.Ck / 8;
    const int g0 = (int)((int64_t)groups * wave / 4), g1 = (int)((int64_t)groups * (wave + 1) / 4);
    const float* xr = p.x + (((int64_t)n * p.Hi + (pok ? iy : 0)) * p.Wi + (pok ? ix : 0)) * p.ldx + 4 * h;
    const float* wr = p.w + (int64_t)(o0 + row) * p.w_row + (int64_t)cl.wtap[t] * p.Ck + 4 * h;
    const float* sr = p.in_scale != nullptr ? p.in_scale + (int64_t)n * p.Ck + 4 * h : nullptr;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int g = g0; g < g1; g += CS_BATCH) {
        float4 xa[CS_BATCH], wb[CS_BATCH], sv[CS_BATCH];
#pragma unroll
        for (int j = 0; j < CS_BATCH; ++j) {
            const bool ok = g + j < g1;
            const int kb = (ok ? g + j : g0) * 8;
            xa[j] = (ok && pok) ? *reinterpret_cast<const float4*>(xr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            wb[j] = ok ? *reinterpret_cast<const float4*>(wr + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
            sv[j] = sr != nullptr ? *reinterpret_cast<const float4*>(sr + kb) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int j = 0; j < CS_BATCH; ++j) {
            // lane (row, h) holds channels kb + 4h .. + 3 of its cell / output channel: instruction q contracts the pair (kb + q, kb + 4 + q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].x * sv[j].x, wb[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].y * sv[j].y, wb[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].z * sv[j].z, wb[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j].w * sv[j].w, wb[j].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][row] = acc[r];
    __syncthreads();
    const int er = threadIdx.x >> 3, eq = threadIdx.x & 7;
    const int em = tile * CS_CELLS + er;
    if (em >= cells) return;
    const int ey = em / cl.Wa, ex = em - ey * cl.Wa;
    float* o = p.out + (((int64_t)n * p.Ho + ey * p.out_stride + cl.out_py) * p.Wo + ex * p.out_stride + cl.out_px) * p.ldo + o0 + eq * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        eg3d_acc(o + q, (red[0][er][eq * 4 + q] + red[1][er][eq * 4 + q]) + (red[2][er][eq * 4 + q] + red[3][er][eq * 4 + q]));
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int eg3d_conv2d_small_supported(const eg3d_conv_params* p) {
    if (!p || !p->x || !p->w || !p->out) return 0;
    if (p->N < 1 || p->Hi < 1 || p->Wi < 1 || p->Ho < 1 || p->Wo < 1 || p->Ck < 32 || (p->Ck & 7) || p->Nc < CS_OUT || (p->Nc % CS_OUT)) return 0;
    if ((p->ldx & 3) || p->ldx < p->Ck || (p->ldo & 3) || p->ldo < p->Nc || (p->w_row & 3)) return 0;
    if (p->in_stride < 1 || p->out_stride < 1 || p->ncls < 1 || p->ncls > 4) return 0;
    if (!al16(p->x) || !al16(p->w) || !al16(p->out) || (p->in_scale && !al16(p->in_scale))) return 0;
    int wt_max = 0;
    for (int c = 0; c < p->ncls; ++c) {
        const eg3d_conv_class& k = p->cls[c];
        if (k.ntaps < 1 || k.ntaps > 9 || k.Ha < 1 || k.Wa < 1) return 0;
        if ((k.Ha - 1) * p->out_stride + k.out_py >= p->Ho || (k.Wa - 1) * p->out_stride + k.out_px >= p->Wo || k.out_py < 0 || k.out_px < 0) return 0;
        for (int t = 0; t < k.ntaps; ++t) {
            if (k.wtap[t] < 0) return 0;
            wt_max = std::max(wt_max, k.wtap[t]);
        }
    }
    if ((int64_t)(wt_max + 1) * p->Ck > p->w_row) return 0;
    return 1;
}

// out (pre-zeroed by the caller, as for EG3D_EPI_ATOMIC) += conv(x * in_scale, w) over the tap classes of p; every other epilogue field of p is ignored.
extern "C" int eg3d_conv2d_small_atomic(const eg3d_conv_params* p, void* stream) {
    if (!p || !p->x || !p->w || !p->out) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_small_supported(p)) return EG3D_ERR_UNSUPPORTED;
    int max_tiles = 0;
    for (int c = 0; c < p->ncls; ++c) max_tiles = std::max(max_tiles, eg3d_cdiv((int64_t)p->cls[c].Ha * p->cls[c].Wa, CS_CELLS));
    if ((int64_t)p->N * max_tiles > 0x7fffffff) return EG3D_ERR_UNSUPPORTED;
    const dim3 grid(p->N * max_tiles, p->Nc / CS_OUT, p->ncls * 9);
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, p->out, (int64_t)p->N * p->Ho * p->Wo * p->ldo); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(conv_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, *p, max_tiles);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
