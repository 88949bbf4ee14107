// Stride-2 3x3 TRANSPOSED convolution (the up-sampling layers' conv, training/networks_stylegan2.py:311-330 with up=2 ->
// torch_utils/ops/conv2d_resample.py:114-136 -> F.conv_transpose2d) on pre-split operands, all four output parities in one workgroup.
//
// out[2a + ky, 2b + kx] += in[a, b] * w[ky, kx]  <=>  for the output pixel of parity (py, px) at cell (a, b):
//     z[2a + py, 2b + px] = sum over the taps with ky = py (mod 2), kx = px (mod 2) of in[a - ky/2, b - kx/2] * w[ky, kx]
// i.e. four stride-1 correlations (4 / 2 / 2 / 1 taps) over the SAME 2 x 2 input neighbourhood.  conv_v2.hip can run them as four
// tap classes, but a 1-tap class re-stages a full input halo for 1/9 of the arithmetic (L2 -> LDS traffic of ~35 TB/s at full MFMA rate:
// it measured slower than the loader-split kernel).  Here a workgroup owns an 8 x 32 patch of CELLS (a, b) and 64 output channels and
// keeps the accumulators of all four parities (2 rows x 4 parities x one 32 x 32 tile per wave, 8 waves): the input halo (9 x 33 pixels)
// of a 16-channel chunk is staged ONCE by LDS-DMA and used by all nine taps -- the arithmetic intensity of the 3 x 3 kernel.
//   * per chunk: A halo (4 planes x 5 KB) + the nine 64-channel weight tiles (9 x 4 KB), double-buffered: one s_waitcnt vmcnt(0) +
//     one s_barrier per CHUNK (54 MFMAs per wave), the next chunk's 7 DMA instructions per wave in flight meanwhile;
//   * 115 KB of LDS, 8 waves: one workgroup per CU, two waves per SIMD;
//   * split-K over chunks (EG3D_EPI_ATOMIC) for the layers whose cell grids cannot fill 256 CUs (every backbone up layer);
//   * the epilogue is the plain store (the FIR / noise / bias / activation pass of an up layer follows: eg3d_modconv_epilogue_fwd).
// Operand images as in conv_v2.hip (eg3d_split_activation / eg3d_split_weight); same arithmetic: three fp16 products per fp32 product, or
// the high pieces only (products = 1).
#include "common.h"
#include "det.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int PH = 8, PW = 32, BN = 64;
constexpr int HW = PW + 2;                      // LDS row pitch of the halo (33 columns used)
constexpr int A_PARTS = 5;                      // 64-slot wave-instructions per A plane (9 x 34 = 306 <= 320 slots)
constexpr int APLANE = A_PARTS * 64 * 16;       // 5120
constexpr int ABUF = 4 * APLANE;                // (piece, k-octet) planes of one chunk
constexpr int BPLANE = BN * 16;                 // 1024: one (tap, piece, k-octet) plane = one wave-instruction
constexpr int BTAP = 4 * BPLANE, BBUF = 9 * BTAP;
constexpr int LDS_A = 0, LDS_B = 2 * ABUF;
constexpr int LDS_MAIN = 2 * ABUF + 2 * BBUF;   // 114688
constexpr int LDS_N = BN + 4;                   // epilogue staging row (floats)
constexpr int LDS_EPI = 4 * 64 * LDS_N * 4;     // four wave-row regions of 64 output pixels x 64 channels
constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
constexpr int N_A = 4 * A_PARTS, N_DMA = N_A + 36;   // 56 DMA wave-instructions per chunk = 7 per wave

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, unsigned lds_byte, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 16, voff, 0, 0, 0);
}

template <bool FULL>
__global__ void __launch_bounds__(512, 2) conv_v2_up2_kernel(const eg3d_conv_up2_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_x = (p.Wc + PW - 1) / PW, tiles_y = (p.Hc + PH - 1) / PH, ntile_n = p.Nc / BN;
    const int ntile = p.N * tiles_y * tiles_x * ntile_n;
    int bid = blockIdx.x;
    if (bid >= ntile) return;
    bid = eg3d_xcd_remap(bid, ntile);
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int n = bid / tiles_y;
    const int y0 = ty * PH, x0 = tx * PW, n0 = n_t * BN;
    const int nchunk = p.Ck / 16;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int c0 = (int)((int64_t)blockIdx.y * nchunk / ks), c1 = (int)((int64_t)(blockIdx.y + 1) * nchunk / ks);
    const int planeA = p.Hi * p.Wi * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * planeA), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)9 * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- loader: wave w issues the wave-instructions j = w + 8 i (i = 0..6) of a chunk; j < 20: A plane j / 5, part j % 5; else weight tap
    // (j - 20) / 4, plane (j - 20) % 4.  A instructions only occur at i <= 2.
    unsigned a_pix[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = wave + 8 * i;
        const int slot = (j % A_PARTS) * 64 + lane;
        const int hy = slot / HW, hx = slot - hy * HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = j < N_A && hy <= PH && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        a_pix[i] = ok ? (unsigned)((y * p.Wi + x) * 16) : OOB;
    }
    auto issue_chunk = [&](int chunk, int buf) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int j = wave + 8 * i;
            if (i < 3 && j < N_A) {
                const int plane = j / A_PARTS, part = j % A_PARTS, piece = plane >> 1, koct = plane & 1;
                const unsigned plane_off = (unsigned)((((n * 2 + piece) * (p.Ck / 8)) + chunk * 2 + koct) * planeA);
                const unsigned pix = a_pix[i < 3 ? i : 0];
                glds16(ars, lds0 + LDS_A + buf * ABUF + plane * APLANE + part * 1024, (pix == OOB || (!FULL && piece == 1)) ? OOB : pix + plane_off);
            } else {
                const int jb = j - N_A, tap = jb >> 2, plane = jb & 3;
                const unsigned v = (unsigned)(((((p.wtap[tap] * nchunk + chunk) * 4 + plane) * p.Nc) + n0 + lane) * 16);
                glds16(wrs, lds0 + LDS_B + buf * BBUF + tap * BTAP + plane * BPLANE, (!FULL && plane >= 2) ? OOB : v);
            }
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    const unsigned a_lane = (unsigned)((lane & 31) * 16 + (lane >> 5) * APLANE + (wm * 2 + 1) * HW * 16 + 16);   // halo origin = (y0 - 1, x0 - 1)
    const unsigned b_lane = (unsigned)((wn * 32 + (lane & 31)) * 16 + (lane >> 5) * BPLANE);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    issue_chunk(c0, 0);
    for (int chunk = c0; chunk < c1; ++chunk) {
        const int buf = (chunk - c0) & 1;
        step_sync<0>();                  // common.h: this chunk's tiles have landed; everybody's LDS reads of the previous chunk (the buffer refilled next) have returned
        if (chunk + 1 < c1) issue_chunk(chunk + 1, buf ^ 1);
        const unsigned abuf = LDS_A + buf * ABUF + a_lane, bbuf = LDS_B + buf * BBUF + b_lane;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int par = (ky & 1) * 2 + (kx & 1);
            const int dy = -(ky >> 1), dx = -(kx >> 1);
            const unsigned bbase = bbuf + t * BTAP;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(smem + bbase);
            f16x8 bl, bg;
            if constexpr (FULL) {
                bl = *reinterpret_cast<const f16x8*>(smem + bbase + 2 * BPLANE);
                const f16x2* s2 = reinterpret_cast<const f16x2*>(&bh);
                f16x2* d2 = reinterpret_cast<f16x2*>(&bg);
#pragma unroll
                for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned abase = abuf + (unsigned)(((i + dy) * HW + dx) * 16);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + abase);
                if constexpr (FULL) {
                    const f16x8 al = *reinterpret_cast<const f16x8*>(smem + abase + 2 * APLANE);
                    acc[i][par] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bg, acc[i][par], 0, 0, 0);        // small terms first
                    acc[i][par] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][par], 0, 0, 0);
                }
                acc[i][par] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][par], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- epilogue: per (row of the wave pair, output-row parity) the two column parities interleave into 64 consecutive output pixels;
    // the tile goes through LDS so that every global access is 16 bytes per lane ------------------------------------------------------
    const float out_mul = 1.f / (*p.a_scale * *p.w_scale);              // exact powers of two
    float* stage = reinterpret_cast<float*>(smem);
    if (p.epi == EG3D_EPI_ATOMIC) {
        // split-K partial tile: atomics straight from the accumulator layout, one lane per channel (a wave-instruction = two runs of 32
        // consecutive floats); the staged float4 form spreads each instruction over sixteen cache lines and measured several times slower
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int a = y0 + wm * 2 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (a >= p.Hc || b >= p.Wc) continue;
#pragma unroll
                for (int par = 0; par < 4; ++par) {
                    const int y = 2 * a + (par >> 1), x = 2 * b + (par & 1);
                    if (y >= p.Ho || x >= p.Wo) continue;
                    eg3d_acc(p.out + ((int64_t)(n * p.Ho + y) * p.Wo + x) * p.ldo + n0 + wn * 32 + (lane & 31), acc[i][par][r] * out_mul);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            __syncthreads();
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    stage[(wm * 64 + m * 2 + px) * LDS_N + wn * 32 + (lane & 31)] = acc[i][py * 2 + px][r] * out_mul;
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int u = tid + k * 512;
                const int region = u >> 10, px64 = (u >> 4) & 63, c4 = u & 15;
                const int a = y0 + region * 2 + i, b = x0 + (px64 >> 1);
                const int y = 2 * a + py, x = 2 * x0 + px64;
                if (a >= p.Hc || b >= p.Wc || y >= p.Ho || x >= p.Wo) continue;
                const float4 v = *reinterpret_cast<const float4*>(stage + (region * 64 + px64) * LDS_N + c4 * 4);
                float* o = p.out + ((int64_t)(n * p.Ho + y) * p.Wo + x) * p.ldo + n0 + c4 * 4;
                *reinterpret_cast<float4*>(o) = v;
            }
        }
    }
}

// ---- 4-row form (round 6): 4 waves, tap-row ring, TWO workgroups per CU ----------------------------------------------------------------------------
// The 8-row kernel above keeps 115 KB of LDS (nine weight tiles, double-buffered): one workgroup per CU, so nothing computes while a workgroup runs its
// store-bound epilogue (the output is four times the input: ~35 % of a round on the SR layers; MfmaUtil 27.6 %), and its eight waves move in lock step.
// Here a workgroup owns 4 x 32 cells x 64 channels (4 waves: wm = row pair, wn = 32-channel half; the same 2 rows x 4 parities per wave = 128 accumulator
// registers) and stages the weights per TAP ROW: a step = the three taps of one ky (same output-row parity, same input row offset) = 18 matrix instructions
// per wave, weight tiles in a three-slot ring (3 x 12 KB) as in conv_v2.hip, the halo (5 x 34 pixels, 12 KB) double-buffered per chunk: 60 KB -> two
// workgroups per CU whose main loops and epilogues overlap, and twice the workgroups for the 128^2 -> 256^2 backbone layer (128 -> 256).
// Issue schedule of ONE wave (up2r_sched, the kernel's loops and vmcnt immediates both come from it; checked on the compiled code by isa_protocol.py):
//   prologue: 3 A parts of the first chunk, B(step 0) [3 operations], B(step 1) [3];   step (s = ky, last chunk?): 1 A part of the next chunk, then B(step + 2) [3]
constexpr int R_PH = 4;
constexpr int R_APARTS = 3;                       // 64-slot wave-instructions per A plane (5 x 34 = 170 <= 192 slots)
constexpr int R_APLANE = R_APARTS * 64 * 16;      // 3072
constexpr int R_ABUF = 4 * R_APLANE;              // 12288
constexpr int R_BGRP = 3 * BTAP;                  // 12288: the three taps of one ky
constexpr int R_LDS_A = 0, R_LDS_B = 2 * R_ABUF;
constexpr int R_LDS_MAIN = 2 * R_ABUF + 3 * R_BGRP;          // 61440
constexpr int R_LDS_EPI = 2 * 64 * LDS_N * 4;     // two wave-row regions of 64 output pixels x 64 channels
static_assert(R_LDS_EPI <= R_LDS_MAIN, "epilogue staging fits the main-loop image");
struct up2r_sched {
    static constexpr int n_a(int s, bool last) { return last ? 0 : 1; }
    static constexpr bool b_issued(int s, bool last) { return !last || s + 2 < 3; }
    static constexpr int n_b(int s, bool last) { return b_issued(s, last) ? 3 : 0; }
    // operations that may stay in flight at the boundary in front of step (s, last): what the previous step issued after B(this step) -- its A part (it precedes
    // its B operations) and its B operations -- except at s = 0, where the A parts are this chunk's own halo
    static constexpr int allow(int s, bool last) {
        const int ps = s >= 1 ? s - 1 : 2;
        const bool pl = s >= 1 ? last : false;
        return n_b(ps, pl) + (s == 0 ? 0 : n_a(ps, pl));
    }
};
static_assert(up2r_sched::allow(2, true) == 0 && up2r_sched::allow(0, false) == 3 && up2r_sched::allow(1, false) == 4, "up2r wait schedule");

template <bool FULL>
__global__ void __launch_bounds__(256, 2) conv_v2_up2r_kernel(const eg3d_conv_up2_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_x = (p.Wc + PW - 1) / PW, tiles_y = (p.Hc + R_PH - 1) / R_PH, ntile_n = p.Nc / BN;
    const int ntile = p.N * tiles_y * tiles_x * ntile_n;
    int bid = blockIdx.x;
    if (bid >= ntile) return;
    bid = eg3d_xcd_remap(bid, ntile);
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int n = bid / tiles_y;
    const int y0 = ty * R_PH, x0 = tx * PW, n0 = n_t * BN;
    const int nchunk = p.Ck / 16;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int c0 = (int)((int64_t)blockIdx.y * nchunk / ks), c1 = (int)((int64_t)(blockIdx.y + 1) * nchunk / ks);
    const int planeA = p.Hi * p.Wi * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * planeA), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)9 * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- loaders.  A: wave w issues the wave-instructions j = w + 4 i (i = 0..2) of a chunk: plane j / 3, part j % 3.
    //      B: per step wave w issues plane w of the three taps of a row.
    unsigned a_pix[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = wave + 4 * i;
        const int slot = (j % R_APARTS) * 64 + lane;
        const int hy = slot / HW, hx = slot - hy * HW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = hy <= R_PH && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        a_pix[i] = ok ? (unsigned)((y * p.Wi + x) * 16) : OOB;
    }
    auto issue_A = [&](int chunk, int buf, int i) {
        const int j = wave + 4 * i;
        const int plane = j / R_APARTS, part = j % R_APARTS, piece = plane >> 1, koct = plane & 1;
        const unsigned plane_off = (unsigned)((((n * 2 + piece) * (p.Ck / 8)) + chunk * 2 + koct) * planeA);
        glds16(ars, lds0 + R_LDS_A + buf * R_ABUF + plane * R_APLANE + part * 1024, (a_pix[i] == OOB || (!FULL && piece == 1)) ? OOB : a_pix[i] + plane_off);
    };
    int wtap_r[9];                       // (scalar registers: through `p.` they are re-fetched from the kernel-argument segment after every boundary)
#pragma unroll
    for (int t = 0; t < 9; ++t) wtap_r[t] = __builtin_amdgcn_readfirstlane(p.wtap[t]);
    // (wave w brings plane w -- (piece, k-octet) -- of the three taps kx = 0, 1, 2 of the row: the tap index is a compile-time constant at every call site)
    auto issue_B = [&](int chunk, auto ky_tag, int slot) {
        constexpr int ky = decltype(ky_tag)::value;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const unsigned v = (unsigned)(((((wtap_r[3 * ky + kx] * nchunk + chunk) * 4 + wave) * p.Nc) + n0 + lane) * 16);
            glds16(wrs, lds0 + R_LDS_B + slot * R_BGRP + kx * BTAP + wave * BPLANE, (!FULL && wave >= 2) ? OOB : v);
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    const unsigned a_lane = (unsigned)((lane & 31) * 16 + (lane >> 5) * R_APLANE + (wm * 2 + 1) * HW * 16 + 16);   // halo origin = (y0 - 1, x0 - 1)
    const unsigned b_lane = (unsigned)((wn * 32 + (lane & 31)) * 16 + (lane >> 5) * BPLANE);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    // ---- prologue: A(c0), B(step 0), B(step 1)
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_A(c0, 0, i);
    issue_B(c0, std::integral_constant<int, 0>{}, 0);
    issue_B(c0, std::integral_constant<int, 1>{}, 1);

    int step = 0;
    auto run_chunk = [&](const int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = (chunk - c0) & 1;
        static_for<0, 3>([&](auto s_tag) {
            constexpr int ky = decltype(s_tag)::value;
            step_sync<up2r_sched::allow(ky, LAST)>();          // common.h: this step's tiles have landed, everybody's LDS reads of the previous step have returned
#pragma unroll
            for (int e = 0; e < up2r_sched::n_a(ky, LAST); ++e) issue_A(chunk + 1, buf ^ 1, ky);
            if constexpr (up2r_sched::b_issued(ky, LAST)) issue_B(chunk + (ky + 2) / 3, std::integral_constant<int, (ky + 2) % 3>{}, (step + 2) % 3);
            constexpr int py = ky & 1, dy = -(ky >> 1);
            const unsigned abuf = R_LDS_A + buf * R_ABUF + a_lane, bbuf = R_LDS_B + (step % 3) * R_BGRP + b_lane;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int par = py * 2 + (kx & 1);
                const int dx = -(kx >> 1);
                const unsigned bbase = bbuf + kx * BTAP;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(smem + bbase);
                f16x8 bl, bg;
                if constexpr (FULL) {
                    bl = *reinterpret_cast<const f16x8*>(smem + bbase + 2 * BPLANE);
                    const f16x2* s2 = reinterpret_cast<const f16x2*>(&bh);
                    f16x2* d2 = reinterpret_cast<f16x2*>(&bg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned abase = abuf + (unsigned)(((i + dy) * HW + dx) * 16);
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + abase);
                    if constexpr (FULL) {
                        const f16x8 al = *reinterpret_cast<const f16x8*>(smem + abase + 2 * R_APLANE);
                        acc[i][par] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bg, acc[i][par], 0, 0, 0);        // small terms first
                        acc[i][par] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][par], 0, 0, 0);
                    }
                    acc[i][par] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][par], 0, 0, 0);
                }
            }
            ++step;
        });
    };
    for (int chunk = c0; chunk + 1 < c1; ++chunk) run_chunk(chunk, std::false_type{});
    run_chunk(c1 - 1, std::true_type{});
    step_sync<0>();                      // every LDS-DMA and LDS read of the main loop is over: the epilogue re-uses the dynamic LDS

    const float out_mul = 1.f / (*p.a_scale * *p.w_scale);              // exact powers of two
    float* stage = reinterpret_cast<float*>(smem);
    if (p.epi == EG3D_EPI_ATOMIC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int a = y0 + wm * 2 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = x0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (a >= p.Hc || b >= p.Wc) continue;
#pragma unroll
                for (int par = 0; par < 4; ++par) {
                    const int y = 2 * a + (par >> 1), x = 2 * b + (par & 1);
                    if (y >= p.Ho || x >= p.Wo) continue;
                    eg3d_acc(p.out + ((int64_t)(n * p.Ho + y) * p.Wo + x) * p.ldo + n0 + wn * 32 + (lane & 31), acc[i][par][r] * out_mul);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            __syncthreads();
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    stage[(wm * 64 + m * 2 + px) * LDS_N + wn * 32 + (lane & 31)] = acc[i][py * 2 + px][r] * out_mul;
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {          // 2 regions x 64 output pixels x 16 channel quads = 2048 units
                const int u = tid + k * 256;
                const int region = u >> 10, px64 = (u >> 4) & 63, c4 = u & 15;
                const int a = y0 + region * 2 + i, b = x0 + (px64 >> 1);
                const int y = 2 * a + py, x = 2 * x0 + px64;
                if (a >= p.Hc || b >= p.Wc || y >= p.Ho || x >= p.Wo) continue;
                const float4 v = *reinterpret_cast<const float4*>(stage + (region * 64 + px64) * LDS_N + c4 * 4);
                float* o = p.out + ((int64_t)(n * p.Ho + y) * p.Wo + x) * p.ldo + n0 + c4 * 4;
                *reinterpret_cast<float4*>(o) = v;
            }
        }
    }
}

std::atomic<uint64_t> g_attr_up[4];

}  // namespace

extern "C" int eg3d_conv2d_up2_supported(const eg3d_conv_up2_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_up2_params& p = *pp;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck < 16 || (p.Ck & 15) || p.Nc < BN || (p.Nc % BN) || (p.ldo & 3) || p.ldo < p.Nc) return 0;
    if (p.Hc < 1 || p.Wc < 1 || p.Hc > p.Hi + 1 || p.Wc > p.Wi + 1 || p.Ho > 2 * p.Hi + 1 || p.Wo > 2 * p.Wi + 1 || p.Ho < 1 || p.Wo < 1) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_ATOMIC) return 0;
    if (p.ksplit > 1 && (p.epi != EG3D_EPI_ATOMIC || p.ksplit > p.Ck / 16 || p.ksplit > 65535)) return 0;
    if (p.patch_rows != 0 && p.patch_rows != 8 && p.patch_rows != 4) return 0;
    for (int t = 0; t < 9; ++t) if (p.wtap[t] < 0 || p.wtap[t] >= 9) return 0;
    if ((int64_t)p.N * 2 * (p.Ck / 8) * p.Hi * p.Wi * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)9 * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_up2(const eg3d_conv_up2_params* pp, void* stream) {
    if (!pp || !pp->a || !pp->w || !pp->out || !pp->a_scale || !pp->w_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_up2_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_up2_params& p = *pp;
    if (reinterpret_cast<uintptr_t>(p.out) & 15) return EG3D_ERR_UNSUPPORTED;
    const bool rows4 = p.patch_rows == 4;
    const int tiles = p.N * eg3d_cdiv(p.Hc, rows4 ? R_PH : PH) * eg3d_cdiv(p.Wc, PW) * (p.Nc / BN);
    const dim3 grid(tiles, p.ksplit > 1 ? p.ksplit : 1, 1);
    hipStream_t st = (hipStream_t)stream;
    EG3D_DET_SCOPE(det, stream);
    if (p.epi == EG3D_EPI_ATOMIC) { EG3D_DET_BIND(det, p.out, (int64_t)p.N * p.Ho * p.Wo * p.ldo); }
    EG3D_DET_COMMIT(det);
    if (rows4) {            // 4 x 32-cell patches, four waves, two workgroups per CU (conv_v2_up2r_kernel)
        if (p.products == 1) {
            auto kern = conv_v2_up2r_kernel<false>;
            if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), R_LDS_MAIN, g_attr_up[3])) return e;
            hipLaunchKernelGGL(kern, grid, dim3(256), R_LDS_MAIN, st, p);
        } else {
            auto kern = conv_v2_up2r_kernel<true>;
            if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), R_LDS_MAIN, g_attr_up[2])) return e;
            hipLaunchKernelGGL(kern, grid, dim3(256), R_LDS_MAIN, st, p);
        }
    } else if (p.products == 1) {
        auto kern = conv_v2_up2_kernel<false>;
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, g_attr_up[1])) return e;
        hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, st, p);
    } else {
        auto kern = conv_v2_up2_kernel<true>;
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, g_attr_up[0])) return e;
        hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, st, p);
    }
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
