// Implicit-GEMM convolution for gfx950, loader-split variant: fp32 operands and results in every arithmetic mode; an fp32 product is
// formed on the matrix cores either exactly (EG3D_PREC_F32: v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) or from 16-bit MFMA products of
// operand pieces cut in the LOADER (F16X3 three fp16 products -- the default of the modulated convs --, BF16X6 six bf16 products,
// BF16X3; include/eg3d_hip.h).  The layers whose grids fill the chip run on the pre-split kernel of conv_v2.hip instead; this kernel
// keeps the transposed / strided classes, split-K, the small grids, 1x1 heads and every non-modulated conv (loss / pose / e4e networks).
//
//   GEMM view:  M = output-grid cells (n,ay,ax)   N = output channels   K = taps x input channels
//   A[m][k]  gathered on the fly from the NHWC activation tensor (im2col never materialised), optionally scaled by the
//            per-(n,k) style (modulation folded into the operand load -> weights are shared by the whole batch);
//   B[n][k]  = w[o][tap][k]  (k contiguous, i.e. the channels_last image of a [O,I,kh,kw] weight).
//
//   Block = 256 threads = 4 waves (WM x WN); block tile BM x BN x (32 fp32 | 16 or 32 split); each wave owns (BM/WM) x (BN/WN) outputs as
//   32x32 MFMA tiles held in 16 accumulator registers each.  Operands go global -> registers -> LDS (k-contiguous rows
//   padded to 36 floats: conflict-free ds_read_b128 for the 4x16-lane service groups of gfx950) and are double buffered:
//   the global loads of step s+1 are in flight while step s is multiplied; one barrier per K-step.
//   One ds_read_b128 per operand row feeds four consecutive MFMAs (the K order inside a tile is permuted identically for A
//   and B, which a GEMM does not care about).
//   Block ids are remapped so that each XCD (private L2) works on a contiguous run of tiles: neighbours share the
//   A halo rows and the weight panel.
//
// Reference semantics being replaced: F.conv2d / F.conv_transpose2d calls of torch_utils/ops/conv2d_resample.py:31-43,
// 114-136 under modulated_conv2d (training/networks_stylegan2.py:34-91), and their autograd data-gradient.
#include "common.h"
#include "det.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 32;                    // K per step, fp32 path
constexpr int LDK = BK + 4;
// split-bf16 paths: K per step is 16 (one v_mfma_f32_32x32x16_bf16 deep) for the 128x128 tile, 32 for the smaller tiles.
// LDS image of one operand tile of R rows: [piece][k-octet = 2*chunk + h][row][8 bf16]; a wave's fragment read (lane = row, h)
// is two contiguous 512-byte runs -> conflict-free ds_read_b128.  The pad staggers the planes for the loader's 8-byte writes (a wave
// writes 8 rows x 16 bytes into each of four planes at KB = 32): measured over all loader-split launches of a C2 step, pad 0: 2239 us,
// 16: 2126, 32: 2097-2111, 48: 2128, 64: 2165, 96: 2120, 128 (rounds 1-2): 2197-2203, 160: 2149, 192: 2128, 224: 2170.
#ifndef EG3D_SPLIT_PAD
#define EG3D_SPLIT_PAD 32
#endif
__host__ __device__ constexpr int split_plane_bytes(int rows) { return rows * 16 + EG3D_SPLIT_PAD; }
__host__ __device__ constexpr int split_tile_bytes(int rows, int np, int kb) { return np * (kb / 8) * split_plane_bytes(rows); }

// Split-bf16 operand pieces (precision 1/2): an fp32 value is cut into bf16 terms by truncation,
//   x = b0 + b1 + b2 + e,  |e| < 2^-24 |x|   (every subtraction below is exact; the pieces are the operands of
//   v_mfma_f32_32x32x16_bf16, whose products are exact and accumulated in fp32).
// loader-side split of four consecutive-k values into NP bf16 pieces (4 bf16 = 8 bytes per piece)
template <int NP>
__device__ __forceinline__ void split4(const float4 v, uint2* out) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned w[3][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned u0 = __float_as_uint(x[2 * q]), u1 = __float_as_uint(x[2 * q + 1]);
        w[0][q] = __builtin_amdgcn_perm(u1, u0, 0x07060302);
        const float r0 = x[2 * q] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * q + 1] - __uint_as_float(u1 & 0xffff0000u);
        const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
        w[1][q] = __builtin_amdgcn_perm(v1, v0, 0x07060302);
        const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
        w[2][q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
    }
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) out[pc] = make_uint2(w[pc][0], w[pc][1]);
}

// Two-piece fp16 split (precision 3): x = h + l,  h = fp16(x) rounded toward zero,  l = fp16(x - h) rounded to nearest.  x - h is exact in
// fp32, |l| < 2^-10 |x|, and the residual |x - h - l| <= max(2^-22 |x|, 2^-25) (the second term where l is a subnormal fp16, i.e. for
// |x| < 2^-4).  l is clamped to the fp16 range so that an out-of-range x saturates instead of producing inf - inf.
__device__ __forceinline__ void split4h(const float4 v, uint2* out) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned hw[2], lw[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x[2 * q], x[2 * q + 1]);
        const float r0 = __builtin_amdgcn_fmed3f(x[2 * q] - (float)h[0], -65504.f, 65504.f);
        const float r1 = __builtin_amdgcn_fmed3f(x[2 * q + 1] - (float)h[1], -65504.f, 65504.f);
        const f16x2_t l = {(_Float16)r0, (_Float16)r1};
        __builtin_memcpy(&hw[q], &h, 4);
        __builtin_memcpy(&lw[q], &l, 4);
    }
    out[0] = make_uint2(hw[0], hw[1]);
    out[1] = make_uint2(lw[0], lw[1]);
}

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// waves per SIMD the register allocation is held to: 3 for the 4-wave split-bf16 tiles whose two loader register sets are small
// (<= 168 VGPRs, 3 x 49 KB of LDS per CU), 2 where the sets are larger, unconstrained for the 8-wave / fp32 256-row variants
__host__ __device__ constexpr int conv_min_waves(int bm, int bn, int waves, int prec, bool vec) {
    const int kb = prec ? (bm * bn >= 128 * 128 ? 16 : 32) : 32;
    const int per_thread = (bm + bn) * (kb / 4) / (waves * 64);          // float4 per thread and register set
    // the scalar-epilogue instantiation (split-K launches: few blocks, latency-bound) keeps 16 rows x 4 side inputs in flight per lane and
    // spills badly at 168 registers: it is held to 2 waves per SIMD instead
    return waves != 4 ? 1 : (prec ? (per_thread <= 4 && vec ? 3 : 2) : (per_thread <= 8 ? 2 : 1));
}

// upfirdn2d.upsample2d at one output pixel for a separable 4-tap filter t (true convolution, padding (2, 1), zero outside):
//   out[2i] = t[3] x[i-1] + t[1] x[i],   out[2i+1] = t[2] x[i] + t[0] x[i+1]   along each axis.  low -> this lane's four channels of pixel (0, 0).
__device__ __forceinline__ float4 fir_up2_at(const float* __restrict__ low, int ld, int Hl, int Wl, int y, int x, const float* t) {
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

template <int BM, int BN, int WM, int WN, int PREC, bool VEC>
__global__ void __launch_bounds__(WM * WN * 64, conv_min_waves(BM, BN, WM * WN, PREC, VEC)) conv_igemm_kernel(const eg3d_conv_params p) {
    constexpr int NT = WM * WN * 64;                        // 4 or 8 waves
    constexpr int NP = PREC == 1 ? 3 : 2;                   // bf16 pieces per operand (split paths)
    constexpr int KB = PREC ? (BM * BN >= 128 * 128 ? 16 : 32) : BK;      // K per step
    constexpr int TPR = KB / 4;                             // loader threads per tile row (one float4 each)
    constexpr int RPP = NT / TPR;                           // tile rows covered by one pass of the loader
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LD = BM >= RPP ? BM / RPP : 1, B_LD = BN >= RPP ? BN / RPP : 1;         // float4 per thread per tile
    static_assert((BM >= RPP ? BM % RPP : RPP % BM) == 0 && (BN >= RPP ? BN % RPP : RPP % BN) == 0, "loader");
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int A_STAGE = PREC ? split_tile_bytes(BM, NP, KB) : BM * LDK * 4;      // bytes per pipeline stage
    constexpr int B_STAGE = PREC ? split_tile_bytes(BN, NP, KB) : BN * LDK * 4;
    constexpr int A_PIECE = (KB / 8) * split_plane_bytes(BM), B_PIECE = (KB / 8) * split_plane_bytes(BN);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const As_b = reinterpret_cast<char*>(smem);               // [2] stages
    char* const Bs_b = As_b + 2 * A_STAGE;
    float* As = smem;
    // the row bookkeeping sits behind whichever is larger: the operand stages or the vector epilogue's staging area (which reuses them)
    constexpr int OPER_BYTES = 2 * (A_STAGE + B_STAGE);
    constexpr int EPI_STAGE_BYTES = VEC ? (3 * BN + 4 + WM * 32 * (BN + 4)) * 4 : 0;
    int* rowpix = reinterpret_cast<int*>(As_b + (OPER_BYTES > EPI_STAGE_BYTES ? OPER_BYTES : EPI_STAGE_BYTES));   // [BM] output pixel index (n*Ho+oy)*Wo+ox, -1 = none
    int* rown = rowpix + BM;                  // [BM] batch index

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int cls_id = blockIdx.z % p.ncls;
    const int kslice = blockIdx.z / p.ncls;
    const eg3d_conv_class& cl = p.cls[cls_id];
    const int Ha = cl.Ha, Wa = cl.Wa, ntaps = cl.ntaps;
    const int HWa = Ha * Wa;
    const int Mc = p.N * HWa;
    const int tiles_n = (p.Nc + BN - 1) / BN;
    const int ntile = ((Mc + BM - 1) / BM) * tiles_n;
    int bid = blockIdx.x;
    if (bid >= ntile) return;
    bid = eg3d_xcd_remap(bid, ntile);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    // ---- per-row bookkeeping -------------------------------------------------------------------------------
    if (tid < BM) {
        int m = m0 + tid;
        int pix = -1, n = 0;
        if (m < Mc) {
            n = m / HWa;
            int rem = m - n * HWa;
            int ay = rem / Wa, ax = rem - ay * Wa;
            pix = (n * p.Ho + ay * p.out_stride + cl.out_py) * p.Wo + ax * p.out_stride + cl.out_px;
        }
        rowpix[tid] = pix;
        rown[tid] = n;
    }
    // ---- F16X3 operand range: bring the A operand to ~2^13..2^14 at its maximum with an exact power of two (p.a_amax: device scalar
    // holding max|A|, e.g. written by the producer of a gradient tensor); the accumulators are rescaled in the epilogue.
    float a_mul = 1.f, a_inv = 1.f;
    if constexpr (PREC == 3 || PREC == 5) {
        if (p.a_amax != nullptr) {
            const float am = *p.a_amax * p.a_amax_mul;
            if (am > 0.f && am < 3.0e38f) {
                int e;
                (void)frexpf(am, &e);                       // am = m * 2^e, m in [0.5, 1)
                e = e > 110 ? 110 : (e < -110 ? -110 : e);
                a_mul = ldexpf(1.f, 14 - e);
                a_inv = ldexpf(1.f, e - 14);
            }
        }
    }
    // ---- operand loaders: branch-free raw buffer loads ----------------------------------------------------------------
    // Every thread fetches A_LD + B_LD 16-byte pieces per K-step.  The per-row part of the address and a 9-bit "tap in bounds"
    // mask are computed once; per step the address is base + (wave-uniform tap/chunk offset), and an out-of-image tap (or
    // a row / channel past the end) is redirected to an out-of-range buffer offset, which the hardware returns as zeros.  No
    // divergent control flow in the K loop, so the loads interleave with the MFMAs of the previous step.
    const int lrow = tid / TPR, col4 = tid % TPR;
    constexpr unsigned OOB = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4), 0x00020000);
    int wt_hi = 9;                                // weight taps addressed by this class (kernels larger than 3x3 come as several classes)
    for (int t = 0; t < ntaps; ++t) wt_hi = max(wt_hi, cl.wtap[t] + 1);
    const int64_t w_bytes = ((int64_t)(p.Nc - 1) * p.w_row + (int64_t)wt_hi * p.Ck) * 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)(w_bytes > 0x7fffffe0 ? 0x7fffffe0 : w_bytes), 0x00020000);
    unsigned a_base[A_LD], a_mask[A_LD];
    int a_n[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        int m = m0 + lrow + RPP * j;
        const bool ok = m < Mc && lrow + RPP * j < BM;
        int mm = ok ? m : 0;
        int n = mm / HWa;
        int rem = mm - n * HWa;
        int ay = rem / Wa;
        const int iy0 = ay * p.in_stride, ix0 = (rem - ay * Wa) * p.in_stride;
        a_n[j] = n;
        a_base[j] = (unsigned)((((int64_t)(n * p.Hi + iy0) * p.Wi + ix0) * p.ldx + col4 * 4) * 4);
        unsigned mask = 0;
        for (int t = 0; t < ntaps; ++t) {
            int iy = iy0 + cl.dy[t], ix = ix0 + cl.dx[t];
            if (ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mask |= 1u << t;
        }
        a_mask[j] = mask;
    }
    unsigned b_base[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        int row = n0 + lrow + RPP * j;
        b_base[j] = (row < p.Nc && lrow + RPP * j < BN) ? (unsigned)(((int64_t)row * p.w_row + col4 * 4) * 4) : OOB;
    }

    const int nchunks = (p.Ck + KB - 1) / KB;
    const int S = nchunks * ntaps;
    const int s_begin = (int)((int64_t)kslice * S / p.ksplit);
    const int s_end = (int)((int64_t)(kslice + 1) * S / p.ksplit);

    // Two register sets: the global loads of step s+2 / s+3 are in flight while step s is multiplied and step s+1 is split and
    // written to LDS, so the split arithmetic never waits on memory and can be interleaved with the MFMAs of the same wave.
    struct Regs { float4 ra[A_LD], rb[B_LD], sc[A_LD]; int chunk; };
    Regs R0, R1;
    R0.chunk = R1.chunk = -1;
    int nx_chunk = s_begin / ntaps, nx_tap = s_begin - nx_chunk * ntaps;      // (chunk, tap) of the next step to load

    auto load_tiles = [&](Regs& R) {
        float4 (&ra)[A_LD] = R.ra; float4 (&rb)[B_LD] = R.rb; float4 (&sc)[A_LD] = R.sc;
        const int chunk = nx_chunk, tap = nx_tap;
        const bool kok = chunk * KB + col4 * 4 < p.Ck;
        if (p.in_scale != nullptr && chunk != R.chunk) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j)
                sc[j] = kok ? *reinterpret_cast<const float4*>(p.in_scale + (int64_t)a_n[j] * p.Ck + chunk * KB + col4 * 4) : make_float4(0, 0, 0, 0);
            R.chunk = chunk;
        }
        const unsigned aoff = (unsigned)(((cl.dy[tap] * p.Wi + cl.dx[tap]) * p.ldx + chunk * KB) * 4);       // wave-uniform
        const unsigned boff = (unsigned)((cl.wtap[tap] * p.Ck + chunk * KB) * 4);
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const bool ok = kok && ((a_mask[j] >> tap) & 1u);
            auto v = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? a_base[j] + aoff : OOB, 0, 0);
            __builtin_memcpy(&ra[j], &v, 16);
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            auto v = __builtin_amdgcn_raw_buffer_load_b128(wrs, kok ? b_base[j] + boff : OOB, 0, 0);
            __builtin_memcpy(&rb[j], &v, 16);
        }
        if (++nx_tap == ntaps) { nx_tap = 0; ++nx_chunk; }
    };
    auto store_tiles = [&](Regs& R, int buf) {
        float4 (&ra)[A_LD] = R.ra; float4 (&rb)[B_LD] = R.rb; float4 (&sc)[A_LD] = R.sc;
        if constexpr (PREC == 0) {
            float* a = reinterpret_cast<float*>(As_b + buf * A_STAGE);
            float* b = reinterpret_cast<float*>(Bs_b + buf * B_STAGE);
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                float4 v = ra[j];
                if (p.in_scale != nullptr) v = f4mul(v, sc[j]);
                if (BM >= RPP || lrow < BM) *reinterpret_cast<float4*>(a + (lrow + RPP * j) * LDK + col4 * 4) = v;
            }
#pragma unroll
            for (int j = 0; j < B_LD; ++j)
                if (BN >= RPP || lrow < BN) *reinterpret_cast<float4*>(b + (lrow + RPP * j) * LDK + col4 * 4) = rb[j];
        } else {
            // thread (row, col4) owns k = 4*col4 .. +3  ->  k-octet col4>>1, 8-byte slot (col4&1) of the row's 16-byte group
            char* a = As_b + buf * A_STAGE + (col4 >> 1) * split_plane_bytes(BM) + (col4 & 1) * 8;
            char* b = Bs_b + buf * B_STAGE + (col4 >> 1) * split_plane_bytes(BN) + (col4 & 1) * 8;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                float4 v = ra[j];
                if (p.in_scale != nullptr) v = f4mul(v, sc[j]);
                if constexpr (PREC == 3 || PREC == 5) { v.x *= a_mul; v.y *= a_mul; v.z *= a_mul; v.w *= a_mul; }
                uint2 pc[NP];
                if constexpr (PREC == 3 || PREC == 5) split4h(v, pc); else split4<NP>(v, pc);
                if (BM >= RPP || lrow < BM) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(a + q * A_PIECE + (lrow + RPP * j) * 16) = pc[q];
                }
            }
#pragma unroll
            for (int j = 0; j < B_LD; ++j) {
                uint2 pc[NP];
                if constexpr (PREC == 5) {             // pre-split weights (eg3d_split_weight_pieces): the 16 bytes ARE the two pieces
                    __builtin_memcpy(&pc[0], &rb[j].x, 8);
                    __builtin_memcpy(&pc[1], &rb[j].z, 8);
                } else if constexpr (PREC == 3) split4h(rb[j], pc); else split4<NP>(rb[j], pc);
                if (BN >= RPP || lrow < BN) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(b + q * B_PIECE + (lrow + RPP * j) * 16) = pc[q];
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        load_tiles(R0);
        store_tiles(R0, 0);
    }
    if (s_begin + 1 < s_end) load_tiles(R1);
    if (s_begin + 2 < s_end) load_tiles(R0);
    __syncthreads();

    const int arow = wm * (TM * 32) + (lane & 31);
    const int brow = wn * (TN * 32) + (lane & 31);
    const int khalf = (lane >> 5) * 4;

    auto compute = [&](const int buf) {
        if constexpr (PREC == 0) {
        const float* a = reinterpret_cast<const float*>(As_b + buf * A_STAGE);
        const float* b = reinterpret_cast<const float*>(Bs_b + buf * B_STAGE);
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a + (arow + i * 32) * LDK + kc * 8 + khalf);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b + (brow + j * 32) * LDK + kc * 8 + khalf);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        } else {
            // split-bf16: PREC 1 = six products (b0b0, b0b1, b1b0, b1b1, b0b2, b2b0; dropped terms < 2^-23), PREC 2 = three
            // (b0b0, b0b1, b1b0; dropped terms < 2^-15).  Small terms are accumulated first.
            const char* a = As_b + buf * A_STAGE + (lane >> 5) * split_plane_bytes(BM) + arow * 16;
            const char* b = Bs_b + buf * B_STAGE + (lane >> 5) * split_plane_bytes(BN) + brow * 16;
#pragma unroll
            for (int kc = 0; kc < KB / 16; ++kc) {
                bf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[q][i] = *reinterpret_cast<const bf16x8*>(a + q * A_PIECE + kc * 2 * split_plane_bytes(BM) + i * 32 * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[q][j] = *reinterpret_cast<const bf16x8*>(b + q * B_PIECE + kc * 2 * split_plane_bytes(BN) + j * 32 * 16);
                }
                // product-major order: the TM*TN accumulators are independent, so consecutive MFMAs never wait on each other
                constexpr int NPROD = PREC == 1 ? 6 : 3;
                constexpr int PA[6] = {0, 1, 0, 1, 2, 0}, PB[6] = {0, 0, 1, 1, 0, 2};      // issued from the back: small terms first
#pragma unroll
                for (int t = NPROD - 1; t >= 0; --t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (PREC == 3 || PREC == 5) {      // the same bits as two fp16 pieces: h*h + h*l + l*h
                                f16x8 ah, bh;
                                __builtin_memcpy(&ah, &af[PA[t]][i], 16);
                                __builtin_memcpy(&bh, &bf[PB[t]][j], 16);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                            } else {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]][i], bf[PB[t]][j], acc[i][j], 0, 0, 0);
                            }
                        }
            }
        }
    };
    // In-order issue: a wave that emits its 24 MFMAs back to back and only then the ~110 VALU of the next tile's split keeps the
    // matrix pipe idle during the second phase, and the co-resident waves run in lockstep, so they do not fill it either (counters:
    // 50 % MFMA busy).  The steady-state step is one straight-line block and the scheduler is told to weave the two streams.
    auto weave = [&]() {
        if constexpr (PREC != 0) {
            constexpr int NMFMA = TM * TN * (PREC == 1 ? 6 : 3) * (KB / 16);
            constexpr int NVALU = (A_LD + (PREC == 5 ? 0 : B_LD)) * (PREC == 1 ? 22 : (PREC == 2 ? 12 : 14)) + A_LD * 4;
            constexpr int PER = (NVALU + NMFMA - 1) / NMFMA;
            __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) * NP * (KB / 16), 0);      // fragment reads
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);                         // a slice of the split arithmetic
            }
            __builtin_amdgcn_sched_group_barrier(0x200, (A_LD + B_LD) * NP, 0);              // LDS writes of the next tile
            __builtin_amdgcn_sched_group_barrier(0x020, A_LD + B_LD, 0);                     // global loads two steps ahead
        }
    };

    int step = s_begin;
    for (; step + 4 < s_end; step += 2) {          // steady state: every load / store below is unconditional
        compute(0); store_tiles(R1, 1); load_tiles(R1); weave();
        __syncthreads();
        compute(1); store_tiles(R0, 0); load_tiles(R0); weave();
        __syncthreads();
    }
    for (; step < s_end; step += 2) {              // drain
        compute(0);
        if (step + 1 < s_end) store_tiles(R1, 1);
        if (step + 3 < s_end) load_tiles(R1);
        __syncthreads();
        if (step + 1 >= s_end) break;
        compute(1);
        if (step + 2 < s_end) store_tiles(R0, 0);
        if (step + 4 < s_end) load_tiles(R0);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------
    const int epi = p.epi;
    const int n_first = m0 / HWa;
    const int m_last = (m0 + BM < Mc ? m0 + BM : Mc) - 1;
    const bool single_n = (m_last / HWa) == n_first;
    float* ds_lds = As;                     // BN floats, reused after the final barrier
    const bool act_on = VEC && epi == EG3D_EPI_BWD_ACT;                  // + the producing layer's activation backward (common.h); host: VEC only
    const bool bwd_like = epi == EG3D_EPI_BWD || act_on;
    const bool do_ds = bwd_like && p.ds != nullptr && p.xin != nullptr;
    float* const ds_out = p.ds_replicas > 1 ? p.ds + (size_t)(blockIdx.x % p.ds_replicas) * p.N * p.Nc : p.ds;
    if (do_ds && single_n) {
        if (tid < BN) ds_lds[tid] = 0.f;
        __syncthreads();
    }
    const float strength = (epi == EG3D_EPI_FWD && p.noise != nullptr) ? *p.noise_strength : 0.f;
    const float act_slope = eg3d_act_pwl_slope(p.act, p.alpha);       // the fused epilogue takes linear / relu / lrelu only (checked on the host)
    const int HWo = p.Ho * p.Wo;

    // ---- vector epilogue: the tile goes through LDS once so that global traffic is 16 bytes per lane and row-contiguous ----------
    // (the accumulator layout gives every lane one float of 16 different rows; written directly that is 64 four-byte accesses per
    //  thread for the output and again for every side input).  Wave-rows are staged 32 rows at a time: WM x 32 x (BN+4) floats.
    // VEC kernels are launched only when conv_vector_epilogue_ok() holds (host side): no split-K atomics, tiles within one image,
    // 16-byte aligned rows and channel counts that are multiples of 4.
    if constexpr (VEC) {
        constexpr int LDS_N = BN + 4;
        constexpr int UPR = BN / 4;                         // float4 units per row
        constexpr int UNITS = WM * 32 * UPR;
        constexpr int UPT = UNITS / NT;                     // units per thread and pass
        static_assert(UNITS % NT == 0 && NT % UPR == 0, "epilogue mapping");
        float* db_lds = smem + BN;                          // column sums of the fused activation backward (dbias, dd) + a scalar
        float* dq_lds = smem + 2 * BN;
        float* sc_lds = smem + 3 * BN;
        float* stage = smem + 3 * BN + 4;                   // after the column-sum rows
        const eg3d_act_bwd& ab = p.act_bwd;
        eg3d_act_bwd_consts abc = {};
        if (act_on) {
            abc = eg3d_act_bwd_setup(ab);
            if (tid < BN) { db_lds[tid] = 0.f; dq_lds[tid] = 0.f; }
            if (tid == 0) sc_lds[0] = 0.f;
        }
        const bool row_sums = act_on && (ab.dnoise != nullptr || ab.dstrength != nullptr);
        const int c4 = tid % UPR;                           // this thread's column group is the same in every unit it handles
        const int col = n0 + c4 * 4;
        const bool cok = col < p.Nc;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), scl4 = make_float4(1.f, 1.f, 1.f, 1.f), dsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float omax = 0.f;
        if (cok && epi == EG3D_EPI_FWD && p.bias != nullptr) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
        if (cok && (epi == EG3D_EPI_FWD || bwd_like) && p.out_scale != nullptr)
            scl4 = *reinterpret_cast<const float4*>(p.out_scale + (int64_t)n_first * p.Nc + col);
        float4 abd4 = make_float4(1.f, 1.f, 1.f, 1.f), abb4 = make_float4(0.f, 0.f, 0.f, 0.f), accb4 = abb4, accd4 = abb4;
        float accs = 0.f;
        if (act_on && ab.d != nullptr) abd4 = *reinterpret_cast<const float4*>(ab.d + (int64_t)n_first * p.Nc + col);
        if (act_on && ab.bias != nullptr) abb4 = *reinterpret_cast<const float4*>(ab.bias + col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __syncthreads();                                // previous pass fully consumed (and the main loop's LDS reads are done)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDS_N + wn * (TN * 32) + j * 32 + (lane & 31)] = acc[i][j][r] * a_inv;
            __syncthreads();
            constexpr int UG = UPT > 4 ? 4 : UPT;            // units in flight per thread (register budget)
#pragma unroll
            for (int ug = 0; ug < UPT; ug += UG) {
            int offs[UG], pixl[UG];
            float4 va[UG], sa[UG], sb[UG];
            float nz[UG];
#pragma unroll
            for (int uu = 0; uu < UG; ++uu) {               // phase 1: all loads
                const int u = uu;
                const int row = (tid + (ug + uu) * NT) / UPR;       // 0 .. WM*32-1  (wave-row = row / 32)
                const int rl = (row >> 5) * (TM * 32) + i * 32 + (row & 31);
                const int pix = rowpix[rl];
                const bool ok = pix >= 0 && cok;
                offs[u] = ok ? pix * p.ldo + col : -1;
                pixl[u] = pix - n_first * HWo;
                va[u] = *reinterpret_cast<const float4*>(stage + row * LDS_N + c4 * 4);
                sa[u] = make_float4(0.f, 0.f, 0.f, 0.f); sb[u] = sa[u]; nz[u] = 0.f;
                if (ok && (epi == EG3D_EPI_FWD || bwd_like) && p.addend != nullptr) {
                    if (epi == EG3D_EPI_FWD && p.addend_up2) {      // the skip image lives at half resolution: 2 x 2 taps of the separable FIR
                        const int yy = pixl[u] / p.Wo, xx = pixl[u] - yy * p.Wo;
                        sa[u] = fir_up2_at(p.addend + (int64_t)n_first * (HWo >> 2) * p.ldo + col, p.ldo, p.Ho >> 1, p.Wo >> 1, yy, xx, p.addend_taps);
                    } else {
                        sa[u] = *reinterpret_cast<const float4*>(p.addend + offs[u]);
                    }
                }
                if (ok && epi == EG3D_EPI_FWD && p.noise != nullptr) nz[u] = p.noise[(int64_t)n_first * p.noise_nstride + pixl[u]];
                if (ok && act_on && ab.noise != nullptr) nz[u] = ab.noise[(int64_t)n_first * ab.noise_nstride + pixl[u]];
                if (ok && (do_ds || act_on)) sb[u] = *reinterpret_cast<const float4*>(p.xin + offs[u]);
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {                  // phase 2: arithmetic + stores
                if (offs[u] < 0) continue;
                float4 v = va[u];
                if (epi == EG3D_EPI_FWD) {
                    const float nzs = nz[u] * strength;
                    float e[4] = {v.x * scl4.x + nzs + bias4.x, v.y * scl4.y + nzs + bias4.y, v.z * scl4.z + nzs + bias4.z, v.w * scl4.w + nzs + bias4.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        e[q] = eg3d_pwl_fwd(e[q], act_slope) * p.gain;
                        if (p.clamp >= 0.f) e[q] = fminf(fmaxf(e[q], -p.clamp), p.clamp);
                    }
                    v = make_float4(e[0] + sa[u].x, e[1] + sa[u].y, e[2] + sa[u].z, e[3] + sa[u].w);
                } else if (bwd_like) {
                    if (do_ds) { dsum4.x += v.x * sb[u].x; dsum4.y += v.y * sb[u].y; dsum4.z += v.z * sb[u].z; dsum4.w += v.w * sb[u].w; }
                    v = make_float4(v.x * scl4.x + sa[u].x, v.y * scl4.y + sa[u].y, v.z * scl4.z + sa[u].z, v.w * scl4.w + sa[u].w);
                    if (act_on) {                   // v = dout of the layer that produced xin: its activation backward, here
                        float cs;
                        v = eg3d_act_bwd_unit(abc, v, sb[u], abd4, abb4, nz[u] * abc.strength, accb4, accd4, cs);
                        if (row_sums) {             // the UPR consecutive lanes of a row (host: Nc % BN == 0, so all of them are here)
                            cs = eg3d_row_group_sum(cs, UPR);
                            if (c4 == 0) {
                                if (ab.dnoise != nullptr) eg3d_acc(ab.dnoise + (int64_t)n_first * ab.dnoise_nstride + pixl[u], cs * abc.strength);
                                accs += cs * nz[u];
                            }
                        }
                    }
                }
                omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                *reinterpret_cast<float4*>(p.out + offs[u]) = v;
            }
            }
        }
        eg3d_commit_amax_block(omax, p.out_amax);           // max|out| for the consumer's operand range
        if (do_ds || act_on) {                              // block-level column sums, then one atomic per column
            if (cok && do_ds) {
                [[maybe_unused]] float* gds = ds_out + (int64_t)n_first * p.Nc + col;
                EG3D_LDS_ACC(&ds_lds[c4 * 4 + 0], gds + 0, dsum4.x); EG3D_LDS_ACC(&ds_lds[c4 * 4 + 1], gds + 1, dsum4.y);
                EG3D_LDS_ACC(&ds_lds[c4 * 4 + 2], gds + 2, dsum4.z); EG3D_LDS_ACC(&ds_lds[c4 * 4 + 3], gds + 3, dsum4.w);
            }
            if (cok && act_on) {
                if (ab.dbias != nullptr) {
                    [[maybe_unused]] float* gdb = ab.dbias + col;
                    EG3D_LDS_ACC(&db_lds[c4 * 4 + 0], gdb + 0, accb4.x); EG3D_LDS_ACC(&db_lds[c4 * 4 + 1], gdb + 1, accb4.y);
                    EG3D_LDS_ACC(&db_lds[c4 * 4 + 2], gdb + 2, accb4.z); EG3D_LDS_ACC(&db_lds[c4 * 4 + 3], gdb + 3, accb4.w);
                }
                if (ab.dd != nullptr) {
                    [[maybe_unused]] float* gdq = ab.dd + (int64_t)n_first * p.Nc + col;
                    EG3D_LDS_ACC(&dq_lds[c4 * 4 + 0], gdq + 0, EG3D_DET_DIV(accd4.x, abd4.x)); EG3D_LDS_ACC(&dq_lds[c4 * 4 + 1], gdq + 1, EG3D_DET_DIV(accd4.y, abd4.y));
                    EG3D_LDS_ACC(&dq_lds[c4 * 4 + 2], gdq + 2, EG3D_DET_DIV(accd4.z, abd4.z)); EG3D_LDS_ACC(&dq_lds[c4 * 4 + 3], gdq + 3, EG3D_DET_DIV(accd4.w, abd4.w));
                }
                if (ab.dstrength != nullptr && accs != 0.f) EG3D_LDS_ACC(sc_lds, ab.dstrength, accs);
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.Nc) {
                if (do_ds) eg3d_acc(ds_out + (int64_t)n_first * p.Nc + n0 + tid, ds_lds[tid]);
                if (act_on && ab.dbias != nullptr) eg3d_acc(ab.dbias + n0 + tid, db_lds[tid]);
                if (act_on && ab.dd != nullptr)         // dL/dd = sum dy * z,  z = (pre - bias - noise) / d
                    eg3d_acc(ab.dd + (int64_t)n_first * p.Nc + n0 + tid, dq_lds[tid] / (ab.d != nullptr ? ab.d[(int64_t)n_first * p.Nc + n0 + tid] : 1.f));
            }
            if (act_on && ab.dstrength != nullptr && tid == 0 && sc_lds[0] != 0.f) eg3d_acc(ab.dstrength, sc_lds[0]);
        }
    } else {
    constexpr int RC = 8;                     // rows per lane whose side inputs are in flight together
    float omax = 0.f;
    // ---- scalar epilogue (split-K atomics, tiles spanning several images, unaligned or odd channel counts) -----------------------

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
        const bool cok = col < p.Nc;
        float dsum = 0.f;
        const float bias = (epi == EG3D_EPI_FWD && p.bias != nullptr && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // Two phases per quarter tile (RC rows per lane): every side input (style scale, noise, skip addend, layer input for the
            // style gradient) is requested first, the arithmetic and the stores follow.  Interleaving them per element makes
            // each load wait for the previous store (the compiler must assume out / addend / xin alias).
#pragma unroll
            for (int rh = 0; rh < 16; rh += RC) {
                int offs[RC];
                float scl[RC], sidea[RC], sideb[RC];
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const int r = rh + q;
                    const int rl = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int pix = rowpix[rl];
                    const bool ok = pix >= 0 && cok;
                    const int off = ok ? pix * p.ldo + col : -1;
                    offs[q] = off;
                    scl[q] = 1.f; sidea[q] = 0.f; sideb[q] = 0.f;
                    if (epi == EG3D_EPI_FWD || epi == EG3D_EPI_BWD) {
                        const int n = rown[rl];
                        if (ok && p.out_scale != nullptr) scl[q] = p.out_scale[(int64_t)n * p.Nc + col];
                        if (ok && p.addend != nullptr) sidea[q] = p.addend[off];
                        if (epi == EG3D_EPI_FWD) {
                            if (ok && p.noise != nullptr) sideb[q] = p.noise[(int64_t)n * p.noise_nstride + (pix - n * HWo)];
                        } else if (ok && do_ds) {
                            sideb[q] = p.xin[off];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const int r = rh + q;
                    const int off = offs[q];
                    if (off < 0) continue;
                    float v = acc[i][j][r] * a_inv;
                    if (epi == EG3D_EPI_STORE) {
                        omax = fmaxf(omax, fabsf(v));
                        p.out[off] = v;
                    } else if (epi == EG3D_EPI_ATOMIC) {
                        eg3d_acc(p.out + off, v);
                    } else if (epi == EG3D_EPI_FWD) {
                        v = v * scl[q] + sideb[q] * strength + bias;
                        v = eg3d_pwl_fwd(v, act_slope) * p.gain;
                        if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
                        omax = fmaxf(omax, fabsf(v + sidea[q]));
                        p.out[off] = v + sidea[q];
                    } else {   // EG3D_EPI_BWD
                        if (do_ds) {
                            const float t = v * sideb[q];
                            if (single_n) dsum += t;
                            else eg3d_acc(ds_out + (int64_t)rown[wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] * p.Nc + col, t);
                        }
                        omax = fmaxf(omax, fabsf(v * scl[q] + sidea[q]));
                        p.out[off] = v * scl[q] + sidea[q];
                    }
                }
            }
        }
        if (do_ds && single_n) {
            dsum += __shfl_xor(dsum, 32);
            if (lane < 32 && cok) EG3D_LDS_ACC(&ds_lds[wn * (TN * 32) + j * 32 + lane], ds_out + (int64_t)n_first * p.Nc + n0 + wn * (TN * 32) + j * 32 + lane, dsum);
        }
    }
    if (do_ds && single_n) {
        __syncthreads();
        if (tid < BN && n0 + tid < p.Nc) eg3d_acc(ds_out + (int64_t)n_first * p.Nc + n0 + tid, ds_lds[tid]);
    }
    if (epi != EG3D_EPI_ATOMIC) eg3d_commit_amax_block(omax, p.out_amax);
    }
}

// host-side test for the vector epilogue (see the kernel): everything the kernel would otherwise have to branch on
template <int BM>
bool conv_vector_epilogue_ok(const eg3d_conv_params& p) {
    if (p.epi == EG3D_EPI_ATOMIC || (p.ldo & 3) || (p.Nc & 3)) return false;     // atomics stay lane-per-column (coalesced per row)
    const void* ptrs[] = {p.out, p.addend, p.xin, p.out_scale, p.bias};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return false;
    if (p.N > 1)                          // every tile must lie within one image (per-image scale / style-gradient rows)
        for (int c = 0; c < p.ncls; ++c)
            if ((p.cls[c].Ha * p.cls[c].Wa) % BM) return false;
    return true;
}

template <int BM, int BN, int WM, int WN, int PREC, bool VEC>
int launch_conv_pv(const eg3d_conv_params& p, hipStream_t st) {
    static std::atomic<uint64_t> attr_done{0};               // one bit per device ordinal
    constexpr int NP = PREC == 1 ? 3 : 2;
    constexpr int KB = BM * BN >= 128 * 128 ? 16 : 32;
    const size_t loop_bytes = PREC ? (size_t)2 * (split_tile_bytes(BM, NP, KB) + split_tile_bytes(BN, NP, KB)) : (size_t)(2 * (BM + BN) * LDK) * sizeof(float);
    const size_t stage_bytes = VEC ? (size_t)(3 * BN + 4 + WM * 32 * (BN + 4)) * sizeof(float) : 0;        // epilogue staging reuses the operand buffers
    const size_t smem = (loop_bytes > stage_bytes ? loop_bytes : stage_bytes) + 2 * BM * sizeof(int);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, PREC, VEC>;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)smem, attr_done)) return e;
    int max_tiles = 0;
    for (int c = 0; c < p.ncls; ++c) {
        int64_t Mc = (int64_t)p.N * p.cls[c].Ha * p.cls[c].Wa;
        int t = eg3d_cdiv(Mc, BM) * eg3d_cdiv(p.Nc, BN);
        if (t > max_tiles) max_tiles = t;
    }
    if (max_tiles == 0) return EG3D_OK;
    dim3 grid(max_tiles, 1, p.ncls * p.ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, p);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

template <int BM, int BN, int WM, int WN, int PREC>
int launch_conv_p(const eg3d_conv_params& p, hipStream_t st) {
    const bool vec = conv_vector_epilogue_ok<BM>(p);
    if (p.addend_up2 && !(vec && p.epi == EG3D_EPI_FWD && p.addend && p.ncls == 1 && p.out_stride == 1 && !(p.Ho & 1) && !(p.Wo & 1) &&
                          p.cls[0].Ha == p.Ho && p.cls[0].Wa == p.Wo))
        return EG3D_ERR_UNSUPPORTED;
    return vec ? launch_conv_pv<BM, BN, WM, WN, PREC, true>(p, st) : launch_conv_pv<BM, BN, WM, WN, PREC, false>(p, st);
}

template <int BM, int BN, int WM, int WN>
int launch_conv(const eg3d_conv_params& p, hipStream_t st) {
    switch (p.precision) {
        case 1: return launch_conv_p<BM, BN, WM, WN, 1>(p, st);
        case 2: return launch_conv_p<BM, BN, WM, WN, 2>(p, st);
        case 3: return p.w_presplit ? launch_conv_p<BM, BN, WM, WN, 5>(p, st) : launch_conv_p<BM, BN, WM, WN, 3>(p, st);
        default: return launch_conv_p<BM, BN, WM, WN, 0>(p, st);
    }
}

// tile configuration: 0 = 128x128 (dominant), 1 = 64x128, 2 = 32x128 (tiny spatial extent), 3 = 128x32 (few output channels),
// 4 = 256x64 (the 64-channel 512^2 layers of the super-resolution head)
int pick_config(const eg3d_conv_params& p) {
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) maxM = std::max<int64_t>(maxM, (int64_t)p.N * p.cls[c].Ha * p.cls[c].Wa);
    if (p.Nc <= 32) return 3;
    if (p.Nc <= 64 && maxM * p.ncls * p.ksplit >= 256 * 256) return 4;
    if (maxM <= 32) return 2;
    // enough 128x128 tiles to fill the chip (2 blocks/CU)?  otherwise shrink the M tile
    int64_t big_tiles = (int64_t)eg3d_cdiv(maxM, 128) * eg3d_cdiv(p.Nc, 128) * p.ncls * p.ksplit;
    if (big_tiles >= 384) return 0;       // (a 256x128 tile with 8 waves was measured slower on every 512^2 / 256^2 layer: 192-227 vs 228-254 TFLOP/s)
    if (maxM <= 64 * 8) return 2;
    return 1;
}

bool act_bwd_ok(const eg3d_conv_params& p) {
    const eg3d_act_bwd& ab = p.act_bwd;
    if (!p.xin || p.ksplit != 1) return false;
    if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return false;            // invertible piecewise-linear activations only
    if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return false;
    if ((reinterpret_cast<uintptr_t>(ab.d) & 15) || (reinterpret_cast<uintptr_t>(ab.bias) & 15)) return false;
    static const int BMs[5] = {128, 64, 32, 128, 256}, BNs[5] = {128, 128, 128, 32, 64};
    const int cfg = pick_config(p);
    if (p.Nc % BNs[cfg]) return false;                                                   // whole rows of channel quads in every tile
    switch (BMs[cfg]) {
        case 128: return conv_vector_epilogue_ok<128>(p);
        case 64: return conv_vector_epilogue_ok<64>(p);
        case 32: return conv_vector_epilogue_ok<32>(p);
        default: return conv_vector_epilogue_ok<256>(p);
    }
}

// Every four consecutive floats of a packed weight matrix -> 16 bytes: their four high fp16 pieces, then the four low pieces -- the
// loader's split4h applied once per weight instead of once per workgroup and K-step (the same bits: results are unchanged).
__global__ void __launch_bounds__(256) split_weight_pieces_kernel(const float4* __restrict__ w, uint4* __restrict__ image, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    uint2 pc[2];
    split4h(w[i], pc);
    image[i] = make_uint4(pc[0].x, pc[0].y, pc[1].x, pc[1].y);
}

}  // namespace

extern "C" int eg3d_split_weight_pieces(const float* w, void* image, int64_t n, void* stream) {
    if (!w || !image || n < 4 || (n & 3) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(image) & 15)) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(split_weight_pieces_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(w), reinterpret_cast<uint4*>(image), n / 4);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_conv2d_igemm_act_bwd_ok(const eg3d_conv_params* pp) {
    return (pp != nullptr && pp->epi == EG3D_EPI_BWD_ACT && pp->ncls >= 1 && pp->ncls <= 4 && pp->ksplit >= 1 && act_bwd_ok(*pp)) ? 1 : 0;
}

extern "C" int eg3d_conv2d_igemm_f32(const eg3d_conv_params* pp, void* stream) {
    if (!pp) return EG3D_ERR_INVALID;
    const eg3d_conv_params& p = *pp;
    if (!p.x || !p.w || !p.out) return EG3D_ERR_INVALID;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck <= 0 || p.Nc <= 0 || p.Ho <= 0 || p.Wo <= 0) return EG3D_ERR_INVALID;
    if (p.ncls < 1 || p.ncls > 4 || p.ksplit < 1 || p.in_stride < 1 || p.out_stride < 1) return EG3D_ERR_INVALID;
    if (p.ksplit > 1 && p.epi != EG3D_EPI_ATOMIC) return EG3D_ERR_INVALID;
    if (p.epi < EG3D_EPI_STORE || p.epi > EG3D_EPI_BWD_ACT) return EG3D_ERR_INVALID;
    if (p.precision < 0 || p.precision > EG3D_PREC_F16X1 || p.ds_replicas < 0) return EG3D_ERR_INVALID;
    if (p.w_presplit && (p.precision != EG3D_PREC_F16X3 || (p.w_row & 3) || (p.Ck & 3))) return EG3D_ERR_INVALID;
    if (p.precision == EG3D_PREC_F16X1) return EG3D_ERR_UNSUPPORTED;      // single-product arithmetic exists in eg3d_conv2d_v2 and the weight gradient; a runtime
                                                                          // skip of the cross products in this kernel's woven main loop measured 4x SLOWER -- callers use F16X3 here
    if (p.epi == EG3D_EPI_FWD && p.noise && !p.noise_strength) return EG3D_ERR_INVALID;
    if (p.epi == EG3D_EPI_FWD && !eg3d_act_is_pwl(p.act)) return EG3D_ERR_UNSUPPORTED;      // other activations: EPI_STORE + eg3d_bias_act
    if ((p.Ck & 3) || (p.ldx & 3) || (p.w_row & 3)) return EG3D_ERR_UNSUPPORTED;   // 16-byte operand loads
    if ((reinterpret_cast<uintptr_t>(p.x) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.in_scale && (reinterpret_cast<uintptr_t>(p.in_scale) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.ldx < p.Ck || p.ldo < p.Nc) return EG3D_ERR_INVALID;
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.Ha <= 0 || k.Wa <= 0 || k.ntaps < 1 || k.ntaps > 9) return EG3D_ERR_INVALID;
        for (int t = 0; t < k.ntaps; ++t)
            if (k.wtap[t] < 0 || (int64_t)(k.wtap[t] + 1) * p.Ck > p.w_row) return EG3D_ERR_INVALID;   // tap outside the weight row
        if ((k.Ha - 1) * p.out_stride + k.out_py >= p.Ho || (k.Wa - 1) * p.out_stride + k.out_px >= p.Wo) return EG3D_ERR_INVALID;
        maxM = std::max<int64_t>(maxM, (int64_t)p.N * k.Ha * k.Wa);
    }
    if ((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4 > 0x7fffffe0ll || (int64_t)p.N * p.Ho * p.Wo * p.ldo > INT32_MAX) return EG3D_ERR_TOO_LARGE;
    if ((int64_t)p.Nc * p.w_row * 4 > 0x7fffffe0ll) return EG3D_ERR_TOO_LARGE;       // 31-bit buffer offsets
    if (p.epi == EG3D_EPI_BWD_ACT && !act_bwd_ok(p)) return EG3D_ERR_UNSUPPORTED;     // callers probe with eg3d_conv2d_igemm_act_bwd_ok
    hipStream_t st = (hipStream_t)stream;
    EG3D_DET_SCOPE(det, stream);
    if (p.epi == EG3D_EPI_ATOMIC) { EG3D_DET_BIND(det, p.out, (int64_t)p.N * p.Ho * p.Wo * p.ldo); }
    EG3D_DET_BIND(det, p.ds, (int64_t)(p.ds_replicas > 1 ? p.ds_replicas : 1) * p.N * p.Nc);
    if (p.epi == EG3D_EPI_BWD_ACT) { EG3D_DET_BIND_ACT(det, p.act_bwd, p.N, p.Nc, (int64_t)p.Ho * p.Wo); }
    EG3D_DET_COMMIT(det);
    int rc;
    switch (pick_config(p)) {
        case 0: rc = launch_conv<128, 128, 2, 2>(p, st); break;
        case 1: rc = launch_conv<64, 128, 2, 2>(p, st); break;
        case 2: rc = launch_conv<32, 128, 1, 4>(p, st); break;
        case 4: rc = launch_conv<256, 64, 4, 1>(p, st); break;
        default: rc = launch_conv<128, 32, 4, 1>(p, st); break;
    }
    EG3D_DET_END(det);
    return rc;
}

extern "C" int eg3d_conv2d_igemm_config(const eg3d_conv_params* pp) {
    if (!pp || pp->ncls < 1 || pp->ncls > 4) return EG3D_ERR_INVALID;
    return pick_config(*pp);
}
