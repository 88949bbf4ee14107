// Weight gradient of a stride-1 k x k convolution from the SPLIT IMAGES of its two operands (pivotal tuning: grads into the generator's
// weights, training/coaches/base_coach.py:96-99; replaces aten::convolution_backward (weight), torch_utils/ops/conv2d_gradfix.py:160-194):
//
//     dw[o, wtap(t), k] += 1 / (gs xs) * sum over (n, y, x) of  G[n, y, x, o] * X[n, y + dy(t), x + dx(t), k]
//
// G = the gradient operand dz, X = the layer input times its styles -- both as the two-piece fp16 images the forward / data-gradient
// kernels of conv_v2.hip already consume ([N][piece 2][C/8][H][W][8] fp16, low piece scaled by 2^11; written by eg3d_split_activation or by
// the fused producers), so nothing is re-read in fp32, re-scaled, rounded or transposed in registers:
//   * GEMM view: M = 64 output channels, N = 64 input channels, K = cells, for ALL taps of the stencil at once (<= 9 accumulator tiles of
//     32 x 32 per wave): a cell of G meets the 3 x 3 neighbourhood of X, so G is fetched once per nine taps and X once per row;
//   * both operands are cell-major in memory ([cell][8 channels]) while the 16-bit MFMA wants 8 consecutive K (= cells) per lane.  The
//     loader-split kernel (conv_wgrad.hip) transposed 4 x 4 blocks in registers; here the images go to LDS as they are, by LDS-DMA, and the
//     fragments come out of gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group fetches a [4 cells][16 channels] block and
//     every lane receives the four cells of its channel) -- no VALU work per element at all;
//   * a workgroup walks a strip of 32 columns down its rows: per row one row of G (32 cells) and ONE new row of X (34 cells: the halo) arrive,
//     two rows ahead of their use (rings of 3 / 5 row slots, counted s_waitcnt vmcnt), 54 (three-product) or 18 (single-product) MFMAs per wave
//     and row;
//   * plane pitch 36 cells = 576 bytes: the two channel octets a 16-lane group reads lie 64 bytes apart modulo the 256-byte bank row.
// Partial tiles (one per workgroup) are added to dw with fp32 atomics, 64 consecutive floats per wave-instruction.
#include "conv_v2_common.h"

namespace {

typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int WG_XW = 32;                     // columns of a strip
constexpr int WG_PITCH = 36;                  // cells per (piece, octet) plane of a row slot
constexpr int WG_PLANE = WG_PITCH * 16;       // 576 B
constexpr int WG_TO = 64, WG_TK = 64;         // channel tile
constexpr int WG_P = 2;                       // rows of look-ahead
constexpr int WG_GR = WG_P + 1, WG_XR = WG_P + 3;

template <int N>
__device__ __forceinline__ void wg_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ f16x8 tr8(const char* smem, unsigned byte) {
    // 8 consecutive cells of this lane's channel: two transposing reads of 4 cells (the second 4 cells = 64 bytes further on)
    auto p0 = (__attribute__((address_space(3))) hv4*)(uintptr_t)((unsigned)(uintptr_t)smem + byte);
    auto p1 = (__attribute__((address_space(3))) hv4*)(uintptr_t)((unsigned)(uintptr_t)smem + byte + 64);
    const f16x4 a = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(p0));
    const f16x4 b = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(p1));
    return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// PIECES: 2 = three products per fp32 product, 1 = high pieces only (EG3D_PREC_F16X1: the SR head of pivotal tuning)
template <int PIECES, int NT>
__global__ void __launch_bounds__(256, 2) conv_wgrad_v2_kernel(const eg3d_wgrad_v2_params p, const int tiles, const int strips_x, const int row_groups) {
    constexpr int NPL = PIECES * 8;                            // planes of a row slot (piece, octet)
    constexpr int NI = (NPL * WG_PITCH + 63) / 64;             // DMA instructions per row slot (1 KB each)
    constexpr int SLOT = NI * 1024;                            // bytes of one row slot (whole instructions: an out-of-range lane still writes its zeros)
    constexpr int NIW = (NI + 3) / 4;                          // ... per wave (uniform: the missing ones fetch nothing, into a dummy KB)
    constexpr int LDS_X = 0, LDS_G = WG_XR * SLOT, LDS_DUMMY = (WG_XR + WG_GR) * SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // tiles fastest: the workgroups of one strip (same cells of G and X, different channel tiles) are neighbours on one XCD
    int L = eg3d_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = L % tiles; L /= tiles;
    const int sx = L % strips_x; L /= strips_x;
    const int rg = L % row_groups;
    const int n = L / row_groups;
    const int tiles_k = p.Ci / WG_TK;
    const int to = tile / tiles_k, tk = tile - to * tiles_k;
    const int H = p.H, W = p.W;
    const int rows_per = (H + row_groups - 1) / row_groups;
    const int y0 = rg * rows_per, y1 = min(H, y0 + rows_per);
    if (y0 >= y1) return;
    const int x0 = sx * WG_XW;
    const int R = y1 - y0;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    constexpr unsigned OOB = 0x7ffffff0u;
    const int plane_bytes = H * W * 16;                                  // one (piece, octet) plane of an image
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0, (int)((int64_t)p.N * 2 * (p.Co / 8) * plane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)((int64_t)p.N * 2 * (p.Ci / 8) * plane_bytes), 0x00020000);

    // ---- DMA items: instruction i of this wave covers the slots (wave + 4 i) 64 + lane of a row slot; slot = plane * 36 + cell -------------
    unsigned g_off[NIW], x_off[NIW];            // byte offset inside the image for row 0 (OOB: nothing to fetch)
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int s = (wave + 4 * i) * 64 + lane;
        const int pl = s / WG_PITCH, cell = s - pl * WG_PITCH;
        const int piece = pl >> 3, oct = pl & 7;
        const bool live = (wave + 4 * i) < NI && pl < NPL;
        const int gx = x0 + cell, xx = x0 + cell - 1;
        g_off[i] = (live && cell < WG_XW && gx < W) ? (unsigned)((((n * 2 + piece) * (p.Co / 8) + to * 8 + oct) * H) * (W * 16) + gx * 16) : OOB;
        x_off[i] = (live && cell < WG_XW + 2 && (unsigned)xx < (unsigned)W) ? (unsigned)((((n * 2 + piece) * (p.Ci / 8) + tk * 8 + oct) * H) * (W * 16) + xx * 16) : OOB;
    }
    auto issue_G = [&](int row, int slot) {          // row: image row (may be >= y1: nothing fetched)
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            glds16(grs, (wave + 4 * i) < NI ? lds0 + LDS_G + slot * SLOT + (wave + 4 * i) * 1024 : lds0 + LDS_DUMMY,
                   (g_off[i] == OOB || row >= y1) ? OOB : g_off[i] + (unsigned)(row * W * 16));
    };
    auto issue_X = [&](int row, int slot) {          // row may be -1 or H: zeros
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            glds16(xrs, (wave + 4 * i) < NI ? lds0 + LDS_X + slot * SLOT + (wave + 4 * i) * 1024 : lds0 + LDS_DUMMY,
                   (x_off[i] == OOB || (unsigned)row >= (unsigned)H) ? OOB : x_off[i] + (unsigned)(row * W * 16));
    };
    // bundle t = { G(y0 + t), X(y0 + t + 1) }: 2 NIW operations per wave.  X row r lives in ring slot (r - y0 + 1) % WG_XR, G row in (r - y0) % WG_GR
    auto issue_bundle = [&](int t) {
        issue_G(y0 + t, t % WG_GR);
        issue_X(y0 + t + 1, (t + 2) % WG_XR);
    };
    issue_X(y0 - 1, 0);
    issue_X(y0, 1);
#pragma unroll
    for (int t = 0; t < WG_P; ++t) issue_bundle(t);

    // ---- fragment addresses: lane l = (kgrp = l >> 5, channel half (l >> 4) & 1, a = l & 15); it SUPPLIES cell a >> 2, channel quad a & 3 of its
    //      group's [4 cells][16 channels] block and RECEIVES the four cells of channel 16 half + a
    const int q = lane & 3;
    const unsigned frag = (unsigned)((2 * ((lane >> 4) & 1) + (q >> 1)) * WG_PLANE + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 16 + (q & 1) * 8);
    const unsigned g_lane = frag + (unsigned)(4 * wm * WG_PLANE);
    const unsigned x_lane = frag + (unsigned)(4 * wn * WG_PLANE);
    constexpr int PIECE_B = 8 * WG_PLANE;                      // (the planes of a slot are contiguous: piece 1 starts 8 planes in)
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int s = 0; s < R; ++s) {
        // bundle s must have landed; younger: bundles s+1 .. s+P-1
        step_sync<(WG_P - 1) * 2 * NIW>();           // common.h: + everybody's LDS reads of the previous row have returned (its ring slots are refilled next)
        issue_bundle(s + WG_P);
        const unsigned gb = LDS_G + (s % WG_GR) * SLOT + g_lane;
#pragma unroll
        for (int ks = 0; ks < WG_XW / 16; ++ks) {
            const f16x8 gh = tr8(smem, gb + ks * 256);
            f16x8 gl, gs;
            if constexpr (PIECES == 2) {
                gl = tr8(smem, gb + ks * 256 + PIECE_B);
                const f16x2* s2 = reinterpret_cast<const f16x2*>(&gh);
                f16x2* d2 = reinterpret_cast<f16x2*>(&gs);
#pragma unroll
                for (int e = 0; e < 4; ++e) d2[e] = s2[e] * k2m11;          // gh 2^-11: the scaled low piece of X meets it
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int dy = p.dy[t], dx = p.dx[t];
                const unsigned xb = LDS_X + ((s + 1 + dy) % WG_XR) * SLOT + x_lane + (unsigned)((1 + dx + ks * 16) * 16);
                const f16x8 xh = tr8(smem, xb);
                if constexpr (PIECES == 2) {
                    const f16x8 xl = tr8(smem, xb + PIECE_B);
                    f16x8 xs;
                    const f16x2* s2 = reinterpret_cast<const f16x2*>(&xh);
                    f16x2* d2 = reinterpret_cast<f16x2*>(&xs);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d2[e] = s2[e] * k2m11;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xs, acc[t], 0, 0, 0);       // small terms first
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gs, xl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh, acc[t], 0, 0, 0);
            }
        }
    }
    wg_wait<0>();

    // ---- partial tile -> dw: row i of the MFMA result = output channel, column = input channel; a wave-instruction = 2 x 32 consecutive floats
    const float mul = 1.f / (*p.g_scale * *p.x_scale);
    const int k = tk * WG_TK + wn * 32 + (lane & 31);
    const int64_t slab = p.slabs ? (int64_t)((n * row_groups + rg) * strips_x + sx) * p.Co * p.w_row : 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* dst = p.dw + slab + (int64_t)p.wtap[t] * p.Ci + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = to * WG_TO + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (p.slabs) dst[(int64_t)o * p.w_row] = acc[t][r] * mul;          // this workgroup's own slab: plain stores, summed in slab order later
            else eg3d_acc(dst + (int64_t)o * p.w_row, acc[t][r] * mul);
        }
    }
}

// ---- weight gradient of an UP-SAMPLING layer (stride-2 3x3 transposed conv) from the PARITY-split image of its gradient operand ---------------
//     dw[o, wtap(ky, kx), k] += 1 / (gs xs) * sum over (n, a, b) of  G_p[n, a + (ky >> 1), b + (kx >> 1), o] * X[n, a, b, k],   p = (ky & 1, kx & 1)
// G_p = the four parity images eg3d_fir44_adjoint_split writes for the data gradient (conv_v2_s2adj.hip), X = the layer input times its styles
// as eg3d_split_activation writes it.  The stride-1 kernel above with the roles of the operands exchanged: X is the operand fetched once per
// row and shared by all nine taps, G the shifted one -- from four images.  Per row step a workgroup brings ONE row of X (32 cells) and one
// new row of EACH parity image (33 cells) into LDS, one step ahead (rings of 2 / 3 row slots per image: 126 KB three-product, one workgroup
// per CU; 70 KB single-product); same transposing fragment reads, same nine accumulators, same atomic tail.
constexpr int WU_P = 1;                          // rows of look-ahead
constexpr int WU_XR = WU_P + 1, WU_GR = WU_P + 2;

template <int PIECES>
__global__ void __launch_bounds__(256, 1) conv_wgrad_v2_up_kernel(const eg3d_wgrad_v2_params p, const int tiles, const int strips_x, const int row_groups) {
    constexpr int NT = 9;
    constexpr int NPL = PIECES * 8;
    constexpr int NI = (NPL * WG_PITCH + 63) / 64;
    constexpr int SLOT = NI * 1024;
    constexpr int NIW = (NI + 3) / 4;
    constexpr int LDS_X = 0, LDS_G = WU_XR * SLOT, LDS_DUMMY = (WU_XR + 4 * WU_GR) * SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int L = eg3d_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = L % tiles; L /= tiles;
    const int sx = L % strips_x; L /= strips_x;
    const int rg = L % row_groups;
    const int n = L / row_groups;
    const int tiles_k = p.Ci / WG_TK;
    const int to = tile / tiles_k, tk = tile - to * tiles_k;
    const int H = p.H, W = p.W, Hp = H + 1, Wp = W + 1;            // cells of X; a parity image of G has one row / column more
    const int rows_per = (H + row_groups - 1) / row_groups;
    const int y0 = rg * rows_per, y1 = min(H, y0 + rows_per);
    if (y0 >= y1) return;
    const int x0 = sx * WG_XW;
    const int R = y1 - y0;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    constexpr unsigned OOB = 0x7ffffff0u;
    const int xplane = H * W * 16, gplane = Hp * Wp * 16;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), 0, (int)((int64_t)p.N * 2 * (p.Co / 8) * 4 * gplane), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)((int64_t)p.N * 2 * (p.Ci / 8) * xplane), 0x00020000);
    unsigned g_off[NIW], x_off[NIW];            // byte offset inside the image for row 0 / parity 0 (OOB: nothing to fetch)
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int s = (wave + 4 * i) * 64 + lane;
        const int pl = s / WG_PITCH, cell = s - pl * WG_PITCH;
        const int piece = pl >> 3, oct = pl & 7;
        const bool live = (wave + 4 * i) < NI && pl < NPL;
        const int gx = x0 + cell;
        x_off[i] = (live && cell < WG_XW && gx < W) ? (unsigned)((((n * 2 + piece) * (p.Ci / 8) + tk * 8 + oct) * H) * (W * 16) + gx * 16) : OOB;
        g_off[i] = (live && cell < WG_XW + 1 && gx < Wp) ? (unsigned)(((((n * 2 + piece) * (p.Co / 8) + to * 8 + oct) * 4) * Hp) * (Wp * 16) + gx * 16) : OOB;
    }
    auto issue_X = [&](int row, int slot) {
#pragma unroll
        for (int i = 0; i < NIW; ++i)
            glds16(xrs, (wave + 4 * i) < NI ? lds0 + LDS_X + slot * SLOT + (wave + 4 * i) * 1024 : lds0 + LDS_DUMMY,
                   (x_off[i] == OOB || row >= y1) ? OOB : x_off[i] + (unsigned)(row * W * 16));
    };
    auto issue_G = [&](int row, int slot) {          // row of the parity images (<= H); all four images
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < NIW; ++i)
                glds16(grs, (wave + 4 * i) < NI ? lds0 + LDS_G + (q * WU_GR + slot) * SLOT + (wave + 4 * i) * 1024 : lds0 + LDS_DUMMY,
                       (g_off[i] == OOB || row > y1) ? OOB : g_off[i] + (unsigned)(q * gplane + row * Wp * 16));
    };
    // bundle t = { X(y0 + t), G_*(y0 + t + 1) }: 5 NIW operations per wave.  X row r in ring slot (r - y0) % WU_XR, G row r in (r - y0) % WU_GR
    auto issue_bundle = [&](int t) {
        issue_X(y0 + t, t % WU_XR);
        issue_G(y0 + t + 1, (t + 1) % WU_GR);
    };
    issue_G(y0, 0);
#pragma unroll
    for (int t = 0; t < WU_P; ++t) issue_bundle(t);

    const int q = lane & 3;
    const unsigned frag = (unsigned)((2 * ((lane >> 4) & 1) + (q >> 1)) * WG_PLANE + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 16 + (q & 1) * 8);
    const unsigned g_lane = frag + (unsigned)(4 * wm * WG_PLANE);
    const unsigned x_lane = frag + (unsigned)(4 * wn * WG_PLANE);
    constexpr int PIECE_B = 8 * WG_PLANE;
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int s = 0; s < R; ++s) {
        step_sync<(WU_P - 1) * 5 * NIW>();
        issue_bundle(s + WU_P);
        const unsigned xb = LDS_X + (s % WU_XR) * SLOT + x_lane;
#pragma unroll
        for (int ks = 0; ks < WG_XW / 16; ++ks) {
            const f16x8 xh = tr8(smem, xb + ks * 256);
            f16x8 xl, xs;
            if constexpr (PIECES == 2) {
                xl = tr8(smem, xb + ks * 256 + PIECE_B);
                const f16x2* s2 = reinterpret_cast<const f16x2*>(&xh);
                f16x2* d2 = reinterpret_cast<f16x2*>(&xs);
#pragma unroll
                for (int e = 0; e < 4; ++e) d2[e] = s2[e] * k2m11;          // xh 2^-11: the scaled low piece of G meets it
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ky = t / 3, kx = t % 3;
                const int par = (ky & 1) * 2 + (kx & 1), sy = ky >> 1, sxx = kx >> 1;
                const unsigned gb = LDS_G + (par * WU_GR + (s + sy) % WU_GR) * SLOT + g_lane + (unsigned)((sxx + ks * 16) * 16);
                const f16x8 gh = tr8(smem, gb);
                if constexpr (PIECES == 2) {
                    const f16x8 gl = tr8(smem, gb + PIECE_B);
                    f16x8 gs;
                    const f16x2* s2 = reinterpret_cast<const f16x2*>(&gh);
                    f16x2* d2 = reinterpret_cast<f16x2*>(&gs);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d2[e] = s2[e] * k2m11;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xs, acc[t], 0, 0, 0);       // small terms first
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gs, xl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh, acc[t], 0, 0, 0);
            }
        }
    }
    wg_wait<0>();

    const float mul = 1.f / (*p.g_scale * *p.x_scale);
    const int k = tk * WG_TK + wn * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* dst = p.dw + (int64_t)p.wtap[t] * p.Ci + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = to * WG_TO + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            eg3d_acc(dst + (int64_t)o * p.w_row, acc[t][r] * mul);
        }
    }
}

std::atomic<uint64_t> g_wg_attr[6];

template <int PIECES, int NT>
int launch_wg(const eg3d_wgrad_v2_params& p, int row_groups, hipStream_t st, int slot) {
    auto kern = conv_wgrad_v2_kernel<PIECES, NT>;
    const int lds = (WG_XR + WG_GR) * ((PIECES * 8 * WG_PITCH + 63) / 64) * 1024 + 1024;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, g_wg_attr[slot])) return e;
    const int tiles = (p.Co / WG_TO) * (p.Ci / WG_TK), strips_x = eg3d_cdiv(p.W, WG_XW);
    const int64_t blocks = (int64_t)tiles * strips_x * row_groups * p.N;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, p, tiles, strips_x, row_groups);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

}  // namespace

extern "C" int eg3d_conv2d_wgrad_v2_supported(const eg3d_wgrad_v2_params* pp) {
    if (!pp) return 0;
    const eg3d_wgrad_v2_params& p = *pp;
    if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.Co < WG_TO || (p.Co % WG_TO) || p.Ci < WG_TK || (p.Ci % WG_TK)) return 0;
    if (p.ntaps != 9 && p.ntaps != 1) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    for (int t = 0; t < p.ntaps; ++t)
        if (p.dy[t] < -1 || p.dy[t] > 1 || p.dx[t] < -1 || p.dx[t] > 1 || p.wtap[t] < 0) return 0;
    if ((int64_t)p.N * 2 * (p.Co / 8) * p.H * p.W * 16 > 0x7fffffe0ll || (int64_t)p.N * 2 * (p.Ci / 8) * p.H * p.W * 16 > 0x7fffffe0ll) return 0;
    return 1;
}

static int wg_row_groups(const eg3d_wgrad_v2_params& p) {
    const int tiles = (p.Co / WG_TO) * (p.Ci / WG_TK), strips_x = eg3d_cdiv(p.W, WG_XW);
    int rg = p.row_groups;
    if (rg <= 0) {              // ~2 workgroups per CU, at least 8 rows each
        const int64_t base = (int64_t)tiles * strips_x * p.N;
        rg = (int)std::max<int64_t>(1, std::min<int64_t>((512 + base - 1) / base, std::max(1, p.H / 8)));
    }
    rg = std::min(rg, p.H);
    return eg3d_cdiv(p.H, eg3d_cdiv(p.H, rg));          // no empty row group: every slab gets written
}

extern "C" int eg3d_conv2d_wgrad_v2_slabs(const eg3d_wgrad_v2_params* pp) {
    if (!pp || !eg3d_conv2d_wgrad_v2_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    return wg_row_groups(*pp) * eg3d_cdiv(pp->W, WG_XW) * pp->N;
}

extern "C" int eg3d_conv2d_wgrad_v2(const eg3d_wgrad_v2_params* pp, void* stream) {
    if (!pp || !pp->g || !pp->x || !pp->dw || !pp->g_scale || !pp->x_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_wgrad_v2_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_wgrad_v2_params& p = *pp;
    const int rg = wg_row_groups(p);
    hipStream_t st = (hipStream_t)stream;
    const bool one = p.products == 1;
    EG3D_DET_SCOPE(det, stream);
    if (!p.slabs) { EG3D_DET_BIND(det, p.dw, (int64_t)p.Co * p.w_row); }
    EG3D_DET_COMMIT(det);
    int rc;
    if (p.ntaps == 9) rc = one ? launch_wg<1, 9>(p, rg, st, 0) : launch_wg<2, 9>(p, rg, st, 1);
    else rc = one ? launch_wg<1, 1>(p, rg, st, 2) : launch_wg<2, 1>(p, rg, st, 3);
    EG3D_DET_END(det);
    return rc;
}

/* up-sampling layers: p->g = parity-split image of G ([N][2][Co/8][4][H + 1][W + 1][8]), p->x = split image of the layer input ([N][2][Ci/8][H][W][8]),
 * H x W = the layer's INPUT resolution, wtap[3 ky + kx] = weight tap of (ky, kx); dy / dx are ignored */
extern "C" int eg3d_conv2d_wgrad_v2_up_supported(const eg3d_wgrad_v2_params* pp) {
    if (!pp) return 0;
    const eg3d_wgrad_v2_params& p = *pp;
    if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.Co < WG_TO || (p.Co % WG_TO) || p.Ci < WG_TK || (p.Ci % WG_TK)) return 0;
    if (p.ntaps != 9 || p.slabs) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    for (int t = 0; t < 9; ++t) if (p.wtap[t] < 0) return 0;
    if ((int64_t)p.N * 2 * (p.Co / 8) * 4 * (p.H + 1) * (p.W + 1) * 16 > 0x7fffffe0ll || (int64_t)p.N * 2 * (p.Ci / 8) * p.H * p.W * 16 > 0x7fffffe0ll) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_wgrad_v2_up(const eg3d_wgrad_v2_params* pp, void* stream) {
    if (!pp || !pp->g || !pp->x || !pp->dw || !pp->g_scale || !pp->x_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_wgrad_v2_up_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_wgrad_v2_params& p = *pp;
    const int tiles = (p.Co / WG_TO) * (p.Ci / WG_TK), strips_x = eg3d_cdiv(p.W, WG_XW);
    int rg = p.row_groups;
    if (rg <= 0) {              // ~1 workgroup per CU (126 KB of LDS), at least 8 rows each
        const int64_t base = (int64_t)tiles * strips_x * p.N;
        rg = (int)std::max<int64_t>(1, std::min<int64_t>((256 + base - 1) / base, std::max(1, p.H / 8)));
    }
    rg = std::min(rg, p.H);
    rg = eg3d_cdiv(p.H, eg3d_cdiv(p.H, rg));
    hipStream_t st = (hipStream_t)stream;
    const bool one = p.products == 1;
    EG3D_DET_SCOPE(det, stream);
    EG3D_DET_BIND(det, p.dw, (int64_t)p.Co * p.w_row);
    EG3D_DET_COMMIT(det);
    const int64_t blocks = (int64_t)tiles * strips_x * rg * p.N;
    if (one) {
        const int lds = (WU_XR + 4 * WU_GR) * ((8 * WG_PITCH + 63) / 64) * 1024 + 1024;
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad_v2_up_kernel<1>), lds, g_wg_attr[4])) return e;
        hipLaunchKernelGGL(conv_wgrad_v2_up_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, st, p, tiles, strips_x, rg);
    } else {
        const int lds = (WU_XR + 4 * WU_GR) * ((16 * WG_PITCH + 63) / 64) * 1024 + 1024;
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad_v2_up_kernel<2>), lds, g_wg_attr[5])) return e;
        hipLaunchKernelGGL(conv_wgrad_v2_up_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, st, p, tiles, strips_x, rg);
    }
    EG3D_LAUNCH_CHECK();
    EG3D_DET_END(det);
    return EG3D_OK;
}
