// Weight-gradient of the implicit-GEMM convolution on the fp32 matrix cores of gfx950.
//
//   dw[o, wtap(t), k] += sum over cells m=(n,ay,ax) of  g[n, ay*os+py, ax*os+px, o] * in_scale[n,k] * x[n, ay*is+dy(t), ax*is+dx(t), k]
//
//   GEMM view per tap: M = output channels (o), N = input channels (k), K = cells.  Both operands are read in their natural
//   NHWC form: a cell contributes one contiguous row of g (o fastest) and one contiguous row of x (k fastest), i.e. the
//   reduction index is the LDS row and the MFMA operands (A[i][kk], B[kk][j] with i/j on lanes 0..31) are conflict-free
//   ds_read_b32 of consecutive addresses.  Block tile 128(o) x 128(k) x 32 cells, 4 waves x (64 x 64), double-buffered
//   through registers like the forward kernel.  The cell range is split over `psplit` blocks per (tile, tap); partial
//   tiles are accumulated into dw with fp32 atomics (dw pre-zeroed by the caller).
//
// Replaces autograd's aten::convolution_backward (weight) for the PTI phase (training/coaches/base_coach.py:96-99,
// torch_utils/ops/conv2d_gradfix.py:166-173).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BC = 32;             // cells per K-step
constexpr int BO = 128, BI = 128;  // tile
constexpr int LDO = BO + 4, LDI = BI + 4;

__global__ void __launch_bounds__(256) conv_wgrad_kernel(const eg3d_wgrad_params p, int tiles_o, int tiles_i, int ntap_total) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Gs = smem;                     // [2][BC][LDO]
    float* Xs = smem + 2 * BC * LDO;      // [2][BC][LDI]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // blockIdx.x -> (tile_o, tile_i); blockIdx.y -> global tap index (class, tap); blockIdx.z -> cell slice
    const int to = blockIdx.x / tiles_i, ti = blockIdx.x % tiles_i;
    int cls_id = 0, tap = blockIdx.y;
    while (cls_id < p.ncls && tap >= p.cls[cls_id].ntaps) { tap -= p.cls[cls_id].ntaps; ++cls_id; }
    if (cls_id >= p.ncls) return;
    const eg3d_conv_class& cl = p.cls[cls_id];
    const int Ha = cl.Ha, Wa = cl.Wa, HWa = Ha * Wa;
    const int Mc = p.N * HWa;
    const int dy = cl.dy[tap], dx = cl.dx[tap], wt = cl.wtap[tap];
    const int o0 = to * BO, k0 = ti * BI;

    const int nsteps_total = (Mc + BC - 1) / BC;
    const int s_begin = (int)((int64_t)blockIdx.z * nsteps_total / p.psplit);
    const int s_end = (int)((int64_t)(blockIdx.z + 1) * nsteps_total / p.psplit);
    if (s_begin >= s_end) return;

    // ---- loaders: branch-free raw buffer loads (as in conv_igemm.hip) -------------------------------------------------------
    // 32 cells x 32 float4 columns per operand and step = 4 + 4 sixteen-byte loads per thread.  A cell outside the range, a tap
    // outside the image or a channel past the end is redirected to an out-of-range buffer offset, which the hardware returns as
    // zeros: no divergent control flow, so the loads and the index arithmetic can be woven between the MFMAs.
    const int lcol = tid & 31, lrow0 = tid >> 5;          // rows lrow0 + 8*j
    constexpr unsigned OOB = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, (int)((int64_t)p.N * p.Ho * p.Wo * p.ldg * 4), 0x00020000);
    const bool ocok = o0 + lcol * 4 < p.Nc, kcok = k0 + lcol * 4 < p.Ck;
    struct Regs { float4 rg[4], rx[4]; };
    Regs R0, R1;
    float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
    int sv_n = -1;

    auto load = [&](Regs& R, int step) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = step * BC + lrow0 + 8 * j;
            const bool ok = m < Mc;
            const int mm = ok ? m : 0;
            const int n = mm / HWa;
            const int rem = mm - n * HWa;
            const int ay = rem / Wa, ax = rem - ay * Wa;
            const unsigned goff = (unsigned)((((int64_t)(n * p.Ho + ay * p.out_stride + cl.out_py) * p.Wo + ax * p.out_stride + cl.out_px) * p.ldg + o0 + lcol * 4) * 4);
            const int iy = ay * p.in_stride + dy, ix = ax * p.in_stride + dx;
            const bool xin = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const unsigned xoff = (unsigned)((((int64_t)(n * p.Hi + iy) * p.Wi + ix) * p.ldx + k0 + lcol * 4) * 4);
            auto gv = __builtin_amdgcn_raw_buffer_load_b128(grs, (ok && ocok) ? goff : OOB, 0, 0);
            auto xv = __builtin_amdgcn_raw_buffer_load_b128(xrs, (ok && kcok && xin) ? xoff : OOB, 0, 0);
            __builtin_memcpy(&R.rg[j], &gv, 16);
            __builtin_memcpy(&R.rx[j], &xv, 16);
            if (p.in_scale != nullptr && p.N > 1 && ok) {      // batch > 1: the style row follows the cell's sample
                const float4 s4 = kcok ? *reinterpret_cast<const float4*>(p.in_scale + (int64_t)n * p.Ck + k0 + lcol * 4) : make_float4(0, 0, 0, 0);
                R.rx[j].x *= s4.x; R.rx[j].y *= s4.y; R.rx[j].z *= s4.z; R.rx[j].w *= s4.w;
            }
        }
    };
    if (p.in_scale != nullptr && p.N == 1 && kcok) sv = *reinterpret_cast<const float4*>(p.in_scale + k0 + lcol * 4);
    (void)sv_n;
    auto store = [&](Regs& R, int buf) {
        float* g = Gs + buf * BC * LDO;
        float* x = Xs + buf * BC * LDI;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(g + (lrow0 + 8 * j) * LDO + lcol * 4) = R.rg[j];
            float4 v = R.rx[j];
            if (p.N == 1) { v.x *= sv.x; v.y *= sv.y; v.z *= sv.z; v.w *= sv.w; }
            *reinterpret_cast<float4*>(x + (lrow0 + 8 * j) * LDI + lcol * 4) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: loads of steps s+2 / s+3 in flight while step s is multiplied and s+1 goes to LDS
    load(R0, s_begin);
    store(R0, 0);
    if (s_begin + 1 < s_end) load(R1, s_begin + 1);
    if (s_begin + 2 < s_end) load(R0, s_begin + 2);
    __syncthreads();
    const int l31 = lane & 31, kh = lane >> 5;
    auto compute = [&](const int buf) {
        const float* g = Gs + buf * BC * LDO + wm * 64 + l31;
        const float* x = Xs + buf * BC * LDI + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BC; kk += 2) {
            const int row = kk + kh;
            float a0 = g[row * LDO], a1 = g[row * LDO + 32];
            float b0 = x[row * LDI], b1 = x[row * LDI + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    // the steady-state step is one straight-line block: weave LDS reads, index arithmetic, global loads and LDS writes between MFMAs
    auto weave = [&]() {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read (64 per step)
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // a slice of the loader arithmetic
            if ((i & 7) == 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // 8 global loads per step
            if ((i & 7) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // 8 LDS writes per step
        }
    };
    int step = s_begin;
    for (; step + 4 < s_end; step += 2) {
        compute(0); store(R1, 1); load(R1, step + 3); weave();
        __syncthreads();
        compute(1); store(R0, 0); load(R0, step + 4); weave();
        __syncthreads();
    }
    for (; step < s_end; step += 2) {
        compute(0);
        if (step + 1 < s_end) store(R1, 1);
        if (step + 3 < s_end) load(R1, step + 3);
        __syncthreads();
        if (step + 1 >= s_end) break;
        compute(1);
        if (step + 2 < s_end) store(R0, 0);
        if (step + 4 < s_end) load(R0, step + 4);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wn * 64 + j * 32 + l31;
            if (k >= p.Ck) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (o < p.Nc) unsafeAtomicAdd(p.dw + (int64_t)o * p.w_row + (int64_t)wt * p.Ck + k, acc[i][j][r]);
            }
        }
}

}  // namespace

extern "C" int eg3d_conv2d_wgrad_f32(const eg3d_wgrad_params* pp, void* stream) {
    if (!pp) return EG3D_ERR_INVALID;
    eg3d_wgrad_params p = *pp;
    if (!p.x || !p.g || !p.dw) return EG3D_ERR_INVALID;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck <= 0 || p.Nc <= 0 || p.Ho <= 0 || p.Wo <= 0) return EG3D_ERR_INVALID;
    if (p.ncls < 1 || p.ncls > 4 || p.in_stride < 1 || p.out_stride < 1) return EG3D_ERR_INVALID;
    if ((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4 > 0x7fffffe0ll || (int64_t)p.N * p.Ho * p.Wo * p.ldg * 4 > 0x7fffffe0ll) return EG3D_ERR_TOO_LARGE;   // 31-bit buffer offsets
    if ((p.Ck & 3) || (p.ldx & 3) || (p.ldg & 3) || p.ldg < ((p.Nc + 3) & ~3)) return EG3D_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(p.x) & 15) || (reinterpret_cast<uintptr_t>(p.g) & 15)) return EG3D_ERR_UNSUPPORTED;
    int ntap_total = 0;
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.Ha <= 0 || k.Wa <= 0 || k.ntaps < 1 || k.ntaps > 9) return EG3D_ERR_INVALID;
        ntap_total += k.ntaps;
        maxM = std::max<int64_t>(maxM, (int64_t)p.N * k.Ha * k.Wa);
    }
    const int tiles_o = eg3d_cdiv(p.Nc, BO), tiles_i = eg3d_cdiv(p.Ck, BI);
    if (p.psplit <= 0) {      // auto: aim for >= ~2048 blocks (measured: 128ch@512^2 81 TF at 1026 blocks, 95 at 1152, flat beyond), at least 8 K-steps per block
        int64_t base = (int64_t)tiles_o * tiles_i * ntap_total;
        int64_t steps = (maxM + BC - 1) / BC;
        int64_t want = (2048 + base - 1) / base;
        p.psplit = (int)std::max<int64_t>(1, std::min<int64_t>(want, std::max<int64_t>(1, steps / 8)));
    }
    static bool attr_done = false;
    const size_t smem = (size_t)(2 * BC * (LDO + LDI)) * sizeof(float);
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid(tiles_o * tiles_i, ntap_total, p.psplit);
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), smem, (hipStream_t)stream, p, tiles_o, tiles_i, ntap_total);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
