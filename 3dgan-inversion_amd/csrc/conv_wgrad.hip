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
#include "det.h"

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

    // logical block id -> (tile_o, tile_i) fastest, then the global tap index (class, tap), then the cell slice.  The hardware deals
    // consecutive workgroups to the 8 XCDs in turn; the remap hands every XCD a contiguous run of logical ids instead, so that all tiles
    // and taps of one cell slice -- which read the same cells of x and g -- run on one XCD and share its L2 (un-mapped, the nine taps of a
    // slice sit on eight different L2s and every one of them fetches its operands from HBM: 2.4 GB for a 128-channel 512^2 layer).
    const int nblk_xy = gridDim.x * gridDim.y;
    const int lid = eg3d_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk_xy * gridDim.z);
    const int bz = lid / nblk_xy, bxy = lid - bz * nblk_xy, by = bxy / gridDim.x, bx = bxy - by * gridDim.x;
    const int to = bx / tiles_i, ti = bx % tiles_i;
    int cls_id = 0, tap = by;
    while (cls_id < p.ncls && tap >= p.cls[cls_id].ntaps) { tap -= p.cls[cls_id].ntaps; ++cls_id; }
    if (cls_id >= p.ncls) return;
    const eg3d_conv_class& cl = p.cls[cls_id];
    const int Ha = cl.Ha, Wa = cl.Wa, HWa = Ha * Wa;
    const int Mc = p.N * HWa;
    const int dy = cl.dy[tap], dx = cl.dx[tap], wt = cl.wtap[tap];
    const int o0 = to * BO, k0 = ti * BI;

    const int nsteps_total = (Mc + BC - 1) / BC;
    const int s_begin = (int)((int64_t)bz * nsteps_total / p.psplit);
    const int s_end = (int)((int64_t)(bz + 1) * nsteps_total / p.psplit);
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
                if (o < p.Nc) eg3d_acc(p.dw + (int64_t)o * p.w_row + (int64_t)wt * p.Ck + k, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Split-arithmetic variant (precision EG3D_PREC_F16X3): the same GEMM on v_mfma_f32_32x32x16_f16 with every fp32 operand cut into
// two fp16 pieces (x = h + l, products hh + hl + lh accumulated in fp32; see conv_igemm.hip), 24 MFMAs of 32 cycles per 32 cells
// instead of 64 MFMAs of 64 cycles.  The 16-bit MFMA wants 8 consecutive reduction indices (cells) per lane, but a cell is a ROW of
// the NHWC operands: a thread therefore loads the same four channels of FOUR CONSECUTIVE cells, transposes them in registers, splits,
// and writes one 8-byte run of four cells per channel and piece into an LDS image laid out [piece][cell octet][channel slot][8 cells]
// -- the fragment layout of conv_igemm.hip.  Channel c of the tile lives in slot (c % 4) * 32 + c / 4, so the 32 lanes that hold the
// same sub-channel write consecutive 16-byte slots (no bank conflicts) and an MFMA tile is 32 slots = channels {4 i + t}.
// g is brought to ~2^13 at its maximum with an exact power of two from the device scalar g_amax (the producer of dz reports it) and
// the accumulators are scaled back; x (activations times styles) is of ordinary magnitude.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

constexpr int H_PLANE = 128 * 16 + 128;           // one cell octet of 128 channel slots (+ pad staggering the two halves of a wave)
constexpr int H_PIECE = 4 * H_PLANE;              // 32 cells = 4 octets
constexpr int H_OPER = 2 * H_PIECE;               // two pieces
constexpr int H_STAGE = 2 * H_OPER;               // g and x

__device__ __forceinline__ void split4h_w(const float a, const float b, const float c, const float d, uint2& hi, uint2& lo) {
    const fp16x2_t h0 = __builtin_amdgcn_cvt_pkrtz(a, b), h1 = __builtin_amdgcn_cvt_pkrtz(c, d);
    const f16x2_t l0 = {(_Float16)__builtin_amdgcn_fmed3f(a - (float)h0[0], -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(b - (float)h0[1], -65504.f, 65504.f)};
    const f16x2_t l1 = {(_Float16)__builtin_amdgcn_fmed3f(c - (float)h1[0], -65504.f, 65504.f), (_Float16)__builtin_amdgcn_fmed3f(d - (float)h1[1], -65504.f, 65504.f)};
    __builtin_memcpy(&hi.x, &h0, 4); __builtin_memcpy(&hi.y, &h1, 4);
    __builtin_memcpy(&lo.x, &l0, 4); __builtin_memcpy(&lo.y, &l1, 4);
}

template <bool ONE>          // ONE: EG3D_PREC_F16X1 -- high pieces only: no low-piece arithmetic, LDS traffic or cross products
__device__ __forceinline__ void conv_wgrad_f16x3_body(const eg3d_wgrad_params& p, float* smem, const int tiles_i, const int bx, const int by, const int bz) {
    char* const lds = reinterpret_cast<char*>(smem);          // [2 stages][g | x][piece][octet][slot][16 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int to = bx / tiles_i, ti = bx % tiles_i;
    int cls_id = 0, tap = by;
    while (cls_id < p.ncls && tap >= p.cls[cls_id].ntaps) { tap -= p.cls[cls_id].ntaps; ++cls_id; }
    if (cls_id >= p.ncls) return;
    const eg3d_conv_class& cl = p.cls[cls_id];
    const int Ha = cl.Ha, Wa = cl.Wa, HWa = Ha * Wa;
    const int Mc = p.N * HWa;
    const int dy = cl.dy[tap], dx = cl.dx[tap], wt = cl.wtap[tap];
    const int o0 = to * BO, k0 = ti * BI;
    const int nsteps_total = (Mc + BC - 1) / BC;
    const int s_begin = (int)((int64_t)bz * nsteps_total / p.psplit);
    const int s_end = (int)((int64_t)(bz + 1) * nsteps_total / p.psplit);
    if (s_begin >= s_end) return;

    float g_mul = 1.f, g_inv = 1.f;
    if (p.g_amax != nullptr) {
        const float am = *p.g_amax * p.g_amax_mul;
        if (am > 0.f && am < 3.0e38f) {
            int e;
            (void)frexpf(am, &e);
            e = e > 110 ? 110 : (e < -110 ? -110 : e);
            g_mul = ldexpf(1.f, 13 - e);
            g_inv = ldexpf(1.f, e - 13);
        }
    }

    const int lcol = tid & 31, lq = tid >> 5;               // channels 4*lcol..+3, cells 4*lq..+3 of the step
    constexpr unsigned OOB = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, (int)((int64_t)p.N * p.Ho * p.Wo * p.ldg * 4), 0x00020000);
    const bool ocok = o0 + lcol * 4 < p.Nc, kcok = k0 + lcol * 4 < p.Ck;
    struct Regs { float4 rg[4], rx[4]; };
    Regs R0, R1;
    float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.in_scale != nullptr && p.N == 1 && kcok) sv = *reinterpret_cast<const float4*>(p.in_scale + k0 + lcol * 4);

    // Cell -> (image, row, column): the steps of a block are loaded strictly in order (s_begin, s_begin + 1, ...), each 32 cells further on.
    // With rows of >= 32 cells and a multiple of four (every layer from 32^2 up) a thread's four cells share a row and the coordinates are
    // advanced incrementally -- the eight integer divisions per step this replaces were several times the MFMA time of the single-product
    // variant.  Other geometries divide as before.
    const bool rows_fast = (Wa % 4 == 0) && Wa >= BC && (HWa % 4 == 0);
    int cn = 0, cay = 0, cax = 0;
    {
        const int m0 = s_begin * BC + 4 * lq;
        cn = m0 / HWa;
        const int rem = m0 - cn * HWa;
        cay = rem / Wa; cax = rem - cay * Wa;
    }
    const unsigned gcol = (unsigned)(p.out_stride * p.ldg), gch = (unsigned)(o0 + lcol * 4), xch = (unsigned)(k0 + lcol * 4);
    auto load = [&](Regs& R, int step) {
        if (rows_fast) {
            // one row base per thread and step, 32-bit element offsets (the launch checks that both tensors stay below 2^31 bytes)
            const int n0 = cn, ay0 = cay, ax0 = cax;
            cax += BC;
            if (cax >= Wa) { cax -= Wa; if (++cay >= Ha) { cay = 0; ++cn; } }
            const bool ok = step * BC + 4 * lq < Mc;                  // Mc is a multiple of four here: the four cells go together
            const int iy = ay0 * p.in_stride + dy;
            const bool yin = ok && kcok && (unsigned)iy < (unsigned)p.Hi;
            const unsigned grow = ((unsigned)(n0 * p.Ho + ay0 * p.out_stride + cl.out_py) * (unsigned)p.Wo + (unsigned)cl.out_px) * (unsigned)p.ldg + gch;
            const unsigned xrow = (unsigned)(n0 * p.Hi + iy) * (unsigned)p.Wi * (unsigned)p.ldx + xch;
            float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (p.in_scale != nullptr && p.N > 1 && ok && kcok) s4 = *reinterpret_cast<const float4*>(p.in_scale + (int64_t)n0 * p.Ck + k0 + lcol * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ax = ax0 + j, ix = ax * p.in_stride + dx;
                const unsigned goff = (grow + (unsigned)ax * gcol) * 4u;
                const unsigned xoff = (xrow + (unsigned)ix * (unsigned)p.ldx) * 4u;
                auto gv = __builtin_amdgcn_raw_buffer_load_b128(grs, (ok && ocok) ? goff : OOB, 0, 0);
                auto xv = __builtin_amdgcn_raw_buffer_load_b128(xrs, (yin && (unsigned)ix < (unsigned)p.Wi) ? xoff : OOB, 0, 0);
                __builtin_memcpy(&R.rg[j], &gv, 16);
                __builtin_memcpy(&R.rx[j], &xv, 16);
                if (p.in_scale != nullptr && p.N > 1) { R.rx[j].x *= s4.x; R.rx[j].y *= s4.y; R.rx[j].z *= s4.z; R.rx[j].w *= s4.w; }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = step * BC + 4 * lq + j;
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
            if (p.in_scale != nullptr && p.N > 1 && ok) {
                const float4 s4 = kcok ? *reinterpret_cast<const float4*>(p.in_scale + (int64_t)n * p.Ck + k0 + lcol * 4) : make_float4(0, 0, 0, 0);
                R.rx[j].x *= s4.x; R.rx[j].y *= s4.y; R.rx[j].z *= s4.z; R.rx[j].w *= s4.w;
            }
        }
    };
    // write: piece planes of the stage; this thread's run = octet lq>>1, bytes (lq&1)*8 of slot (c*32 + lcol)
    auto store = [&](Regs& R, int buf) {
        char* gb = lds + buf * H_STAGE + (lq >> 1) * H_PLANE + lcol * 16 + (lq & 1) * 8;
        char* xb = gb + H_OPER;
        const float gq[4][4] = {{R.rg[0].x, R.rg[1].x, R.rg[2].x, R.rg[3].x}, {R.rg[0].y, R.rg[1].y, R.rg[2].y, R.rg[3].y},
                                {R.rg[0].z, R.rg[1].z, R.rg[2].z, R.rg[3].z}, {R.rg[0].w, R.rg[1].w, R.rg[2].w, R.rg[3].w}};
        const float xq[4][4] = {{R.rx[0].x, R.rx[1].x, R.rx[2].x, R.rx[3].x}, {R.rx[0].y, R.rx[1].y, R.rx[2].y, R.rx[3].y},
                                {R.rx[0].z, R.rx[1].z, R.rx[2].z, R.rx[3].z}, {R.rx[0].w, R.rx[1].w, R.rx[2].w, R.rx[3].w}};
        const float svq[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 hi, lo;
            const float m = p.N == 1 ? svq[c] : 1.f;
            if constexpr (ONE) {
                const fp16x2_t g0 = __builtin_amdgcn_cvt_pkrtz(gq[c][0] * g_mul, gq[c][1] * g_mul), g1 = __builtin_amdgcn_cvt_pkrtz(gq[c][2] * g_mul, gq[c][3] * g_mul);
                __builtin_memcpy(&hi.x, &g0, 4); __builtin_memcpy(&hi.y, &g1, 4);
                *reinterpret_cast<uint2*>(gb + c * 32 * 16) = hi;
                const fp16x2_t x0 = __builtin_amdgcn_cvt_pkrtz(xq[c][0] * m, xq[c][1] * m), x1 = __builtin_amdgcn_cvt_pkrtz(xq[c][2] * m, xq[c][3] * m);
                __builtin_memcpy(&hi.x, &x0, 4); __builtin_memcpy(&hi.y, &x1, 4);
                *reinterpret_cast<uint2*>(xb + c * 32 * 16) = hi;
            } else {
                split4h_w(gq[c][0] * g_mul, gq[c][1] * g_mul, gq[c][2] * g_mul, gq[c][3] * g_mul, hi, lo);
                *reinterpret_cast<uint2*>(gb + c * 32 * 16) = hi;
                *reinterpret_cast<uint2*>(gb + H_PIECE + c * 32 * 16) = lo;
                split4h_w(xq[c][0] * m, xq[c][1] * m, xq[c][2] * m, xq[c][3] * m, hi, lo);
                *reinterpret_cast<uint2*>(xb + c * 32 * 16) = hi;
                *reinterpret_cast<uint2*>(xb + H_PIECE + c * 32 * 16) = lo;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load(R0, s_begin);
    store(R0, 0);
    if (s_begin + 1 < s_end) load(R1, s_begin + 1);
    if (s_begin + 2 < s_end) load(R0, s_begin + 2);
    __syncthreads();
    const int l31 = lane & 31, kh = lane >> 5;
    auto compute = [&](const int buf) {
        const char* g = lds + buf * H_STAGE + kh * H_PLANE + (wm * 64 + l31) * 16;
        const char* x = lds + buf * H_STAGE + H_OPER + kh * H_PLANE + (wn * 64 + l31) * 16;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {                      // two 16-cell MFMA steps per 32-cell stage
            f16x8 a[2][2], b[2][2];                           // [piece][tile]
#pragma unroll
            for (int q = 0; q < (ONE ? 1 : 2); ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[q][t] = *reinterpret_cast<const f16x8*>(g + q * H_PIECE + kc * 2 * H_PLANE + t * 32 * 16);
                    b[q][t] = *reinterpret_cast<const f16x8*>(x + q * H_PIECE + kc * 2 * H_PLANE + t * 32 * 16);
                }
#pragma unroll
            for (int pr = (ONE ? 0 : 2); pr >= 0; --pr) {     // l*h, h*l, then h*h
                const int qa = pr == 2 ? 1 : 0, qb = pr == 1 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[qa][i], b[qb][j], acc[i][j], 0, 0, 0);
            }
        }
    };
    int step = s_begin;
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

    // tile (wm, i) of g holds channels 4*row + (2*wm + i); tile (wn, j) of x holds channels 4*l31 + (2*wn + j).  The 128 x 128 result
    // is put back into natural order through LDS so that the atomics of a wave go to 64 consecutive addresses of one dw row.
    __syncthreads();                                          // all fragment reads of the last stage are done
    float* stage = smem;                                      // [128 o][128 + 4 k] fp32 = 67.6 KB <= 2 * H_STAGE
    constexpr int LDS_K = 128 + 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[(4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + 2 * wm + i) * LDS_K + 4 * l31 + 2 * wn + j] = acc[i][j][r] * g_inv;
    __syncthreads();
    const int kl = tid & 127, k = k0 + kl;
    if (k < p.Ck) {
        float* dst = p.dw + (int64_t)wt * p.Ck + k;
        for (int ol = tid >> 7; ol < 128; ol += 2)
            if (o0 + ol < p.Nc) eg3d_acc(dst + (int64_t)(o0 + ol) * p.w_row, stage[ol * LDS_K + kl]);
    }
}

template <bool ONE>
__global__ void __launch_bounds__(256, 2) conv_wgrad_f16x3_kernel(const eg3d_wgrad_params p, int tiles_o, int tiles_i, int ntap_total) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nblk_xy = gridDim.x * gridDim.y;          // XCD-aware logical block id, as in conv_wgrad_kernel
    const int lid = eg3d_xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nblk_xy * gridDim.z);
    const int bz = lid / nblk_xy, bxy = lid - bz * nblk_xy, by = bxy / gridDim.x, bx = bxy - by * gridDim.x;
    conv_wgrad_f16x3_body<ONE>(p, smem, tiles_i, bx, by, bz);
}

// Several layers in one launch (pivotal tuning queues the weight gradients of its backward pass: the 4^2 .. 64^2 layers and the toRGB
// layers are 10 - 25 us launches of a few dozen workgroups each; together they fill the chip).  grid.x = all blocks, layer by prefix table.
struct WgradBatch {
    eg3d_wgrad_params p[EG3D_WGRAD_BATCH_MAX];
    int blk0[EG3D_WGRAD_BATCH_MAX + 1];
    int nx[EG3D_WGRAD_BATCH_MAX], ny[EG3D_WGRAD_BATCH_MAX], tiles_i[EG3D_WGRAD_BATCH_MAX];
    int n;
};
template <bool ONE>
__global__ void __launch_bounds__(256, 2) conv_wgrad_f16x3_batched_kernel(const WgradBatch b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.blk0[l + 1]) ++l;
    const int nblk = b.blk0[l + 1] - b.blk0[l], nxy = b.nx[l] * b.ny[l];
    const int lid = eg3d_xcd_remap(blockIdx.x - b.blk0[l], nblk);
    const int bz = lid / nxy, bxy = lid - bz * nxy, by = bxy / b.nx[l], bx = bxy - by * b.nx[l];
    conv_wgrad_f16x3_body<ONE>(b.p[l], smem, b.tiles_i[l], bx, by, bz);
}

}  // namespace

// argument checks + the choice of the pixel split (p.psplit is filled in); tiles_o / tiles_i / ntap_total = the launch geometry
static int wgrad_prepare(eg3d_wgrad_params& p, int& tiles_o, int& tiles_i, int& ntap_total, bool& f16) {
    if (!p.x || !p.g || !p.dw) return EG3D_ERR_INVALID;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck <= 0 || p.Nc <= 0 || p.Ho <= 0 || p.Wo <= 0) return EG3D_ERR_INVALID;
    if (p.ncls < 1 || p.ncls > 4 || p.in_stride < 1 || p.out_stride < 1) return EG3D_ERR_INVALID;
    if ((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4 > 0x7fffffe0ll || (int64_t)p.N * p.Ho * p.Wo * p.ldg * 4 > 0x7fffffe0ll) return EG3D_ERR_TOO_LARGE;   // 31-bit buffer offsets
    if ((p.Ck & 3) || (p.ldx & 3) || (p.ldg & 3) || p.ldg < ((p.Nc + 3) & ~3)) return EG3D_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(p.x) & 15) || (reinterpret_cast<uintptr_t>(p.g) & 15)) return EG3D_ERR_UNSUPPORTED;
    ntap_total = 0;
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.Ha <= 0 || k.Wa <= 0 || k.ntaps < 1 || k.ntaps > 9) return EG3D_ERR_INVALID;
        ntap_total += k.ntaps;
        maxM = std::max<int64_t>(maxM, (int64_t)p.N * k.Ha * k.Wa);
    }
    tiles_o = eg3d_cdiv(p.Nc, BO); tiles_i = eg3d_cdiv(p.Ck, BI);
    f16 = p.precision == EG3D_PREC_F16X3 || p.precision == EG3D_PREC_F16X1;
    if (p.psplit <= 0 && f16) {      // ~2048 blocks like the fp32 path, but >= 16 K-steps each (the shorter main loop makes the atomic epilogue weigh more)
        int64_t base = (int64_t)tiles_o * tiles_i * ntap_total;
        int64_t steps = (maxM + BC - 1) / BC;
        int64_t want = (2048 + base - 1) / base;
        // (1x1 layers -- a single tap, few tiles -- are memory-bound streams of x: shorter slices so that every CU gets a block)
        p.psplit = (int)std::max<int64_t>(1, std::min<int64_t>(want, std::max<int64_t>(1, steps / (base >= 9 ? 16 : 8))));
        if (base < 9) p.psplit = (int)std::min<int64_t>(p.psplit, std::max<int64_t>(1, 512 / base));
    }
    if (p.psplit <= 0) {      // auto: aim for >= ~2048 blocks (measured: 128ch@512^2 81 TF at 1026 blocks, 95 at 1152, flat beyond), at least 8 K-steps per block
        int64_t base = (int64_t)tiles_o * tiles_i * ntap_total;
        int64_t steps = (maxM + BC - 1) / BC;
        int64_t want = (2048 + base - 1) / base;
        p.psplit = (int)std::max<int64_t>(1, std::min<int64_t>(want, std::max<int64_t>(1, steps / 8)));
    }
    if (p.precision != EG3D_PREC_F32 && !f16) return EG3D_ERR_UNSUPPORTED;
    return EG3D_OK;
}

extern "C" int eg3d_conv2d_wgrad_f32(const eg3d_wgrad_params* pp, void* stream) {
    if (!pp) return EG3D_ERR_INVALID;
    eg3d_wgrad_params p = *pp;
    int tiles_o, tiles_i, ntap_total;
    bool f16;
    if (int rc = wgrad_prepare(p, tiles_o, tiles_i, ntap_total, f16)) return rc;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, p.dw, (int64_t)p.Nc * p.w_row); EG3D_DET_COMMIT(det);
    if (f16) {
        static std::atomic<uint64_t> attr16{0}, attr16one{0};
        const size_t smem16 = (size_t)2 * H_STAGE;
        const bool one = p.precision == EG3D_PREC_F16X1;
        auto kern = one ? conv_wgrad_f16x3_kernel<true> : conv_wgrad_f16x3_kernel<false>;
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem16, one ? attr16one : attr16)) return e;
        dim3 grid16(tiles_o * tiles_i, ntap_total, p.psplit);
        hipLaunchKernelGGL(kern, grid16, dim3(256), smem16, (hipStream_t)stream, p, tiles_o, tiles_i, ntap_total);
        EG3D_DET_END(det);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
    static std::atomic<uint64_t> attr_done{0};
    const size_t smem = (size_t)(2 * BC * (LDO + LDI)) * sizeof(float);
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad_kernel), (int)smem, attr_done)) return e;
    dim3 grid(tiles_o * tiles_i, ntap_total, p.psplit);
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), smem, (hipStream_t)stream, p, tiles_o, tiles_i, ntap_total);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_conv2d_wgrad_batched(const eg3d_wgrad_params* items, int n, void* stream) {
    if (!items || n < 1) return EG3D_ERR_INVALID;
    static std::atomic<uint64_t> attr16{0}, attr16one{0};
    const size_t smem16 = (size_t)2 * H_STAGE;
    EG3D_DET_SCOPE(det, stream);
    for (int i = 0; i < n; ++i) EG3D_DET_BIND(det, items[i].dw, (int64_t)items[i].Nc * items[i].w_row);
    EG3D_DET_COMMIT(det);
    int i = 0;
    while (i < n) {
        WgradBatch b;
        int blocks = 0, m = 0;
        bool one = false;
        for (; i < n && m < EG3D_WGRAD_BATCH_MAX; ++i, ++m) {
            eg3d_wgrad_params p = items[i];
            int tiles_o, tiles_i, ntap_total;
            bool f16;
            if (int rc = wgrad_prepare(p, tiles_o, tiles_i, ntap_total, f16)) return rc;
            if (!f16) return EG3D_ERR_UNSUPPORTED;                                  // (the fp32 matrix path has no batched form)
            const bool this_one = p.precision == EG3D_PREC_F16X1;
            if (m > 0 && this_one != one) break;                                    // a launch is one instantiation: start a new one
            one = this_one;
            b.p[m] = p; b.blk0[m] = blocks; b.nx[m] = tiles_o * tiles_i; b.ny[m] = ntap_total; b.tiles_i[m] = tiles_i;
            blocks += tiles_o * tiles_i * ntap_total * p.psplit;
        }
        b.blk0[m] = blocks; b.n = m;
        auto kern = one ? conv_wgrad_f16x3_batched_kernel<true> : conv_wgrad_f16x3_batched_kernel<false>;
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem16, one ? attr16one : attr16)) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem16, (hipStream_t)stream, b);
    }
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
