// Weight-streaming split-K convolution for gfx950: the 3x3 stride-1 layers of the 4^2 .. 16^2 blocks (16 .. 256 cells x 512 -> 512 channels at one
// image per GPU; training/networks_stylegan2.py:34-91,417-461 -- forward and data gradient).
//
// At these sizes a launch is 9.4 MB of weights for 0.08 .. 1.2 GFLOP: the only thing to get right is that every weight byte is fetched ONCE, by
// a workgroup that has all its loads in flight before its first matrix instruction.  The loader-split implicit GEMM walks the 4608-deep
// contraction in 36 .. 288 barrier-separated steps of 3 MFMAs per wave (10 .. 30 us per launch, MfmaUtil 2 %, 4.4 x the weight bytes from HBM).
// Here the contraction is cut into its 16-channel chunks ACROSS workgroups and nothing is left to loop over:
//   workgroup = (32-channel tile, group of four chunks, block of <= 256 cells): 16 x 8 x 1 = 128 workgroups for 16^2 x 512 -> 512; wave = chunk;
//   B: the nine taps' weight fragments of the wave's chunk -- 18 KB per wave, each byte of the weight image read by exactly one wave --
//      go straight from global memory into registers (18 x buffer_load_dwordx4 per lane, issued first);
//   A: the chunks' halos (<= 10 x 34 pixels x 16 channels of the fp32 NHWC activation each) are loaded, modulated, range-normalised and split
//      into the two fp16 pieces HERE (a separate operand pass would cost a launch) and parked in LDS as the (piece, k-octet) planes conv_v3.hip reads;
//   one barrier, then 9 taps x (MT x 3) v_mfma_f32_32x32x16_f16 per wave over ALL cells of the block; the four chunk tiles are summed in LDS in
//   wave order and the total is added to the pre-zeroed fp32 output with atomics (EG3D_EPI_ATOMIC's contract: the finishing epilogue is
//   somebody else's launch).  The atomics are what such a launch costs beyond its ~6 us latency chain -- 230 G float adds per second device-wide
//   (one chunk per workgroup, 32 slices of 16^2 x 512: 18 us of a 25 us launch) -- hence four chunks per workgroup, not one.
// Same arithmetic as conv_v2 / conv_v3 (two-piece split, three products, small terms first), so results differ from theirs by summation order only.
#include "conv_v2_common.h"
#include <atomic>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WS_BN = 32;                       // output channels of a workgroup (one MFMA column tile)
constexpr int WS_CPW = 4;                       // 16-channel chunks of the contraction per workgroup = waves
constexpr int WS_MAXSLOTS = 344;                // halo pixels of a cell block: 10 x 34 (32 wide), 18 x 18 (16 wide) ...
constexpr int WS_SPP = 256 / (2 * WS_CPW);      // halo pixels per load pass: thread = (pixel of the pass, chunk, k-octet)

__device__ __forceinline__ f16x8 ws_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// m / d for 0 <= m < 2^20, 1 <= d <= 2^10 without the ~35-instruction integer division sequence (these kernels are latency chains of a few
// microseconds: the halo loaders and the output loops did 40 .. 200 divisions by run-time pitches per lane): float estimate + one correction each way
__device__ __forceinline__ int ws_div(int m, int d, float rcp) {
    int q = (int)((float)m * rcp);
    q -= (q * d > m) ? 1 : 0;
    q += ((q + 1) * d <= m) ? 1 : 0;
    return q;
}

// FULL: three products per fp32 product; !FULL: high pieces only.  MT: MFMA row tiles (32 cells) of the workgroup's cell block (every wave computes
// all of them for ITS chunk).
template <bool FULL, int MT>
__global__ void __launch_bounds__(256) conv_ws_kernel(const eg3d_conv_ws_params p) {
    constexpr int NB = FULL ? 2 : 1;                                  // weight pieces
    constexpr int MB = 32 * MT;
    // halo load passes in flight: the whole halo of the 256-cell block in one global round trip (18 x 18 or 10 x 34 pixels: 11 passes), fewer
    // for the smaller blocks (8 x 8 + halo: 4 passes; more registers and skipped instructions measured +0.7 us there)
    constexpr int WS_BATCH = MT == 8 ? (WS_MAXSLOTS + WS_SPP - 1) / WS_SPP : (MT == 4 ? 7 : 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];       // WS_CPW x 4 planes x SLOTS x 16 bytes; afterwards the [MT][4][64] float4 tile image
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W;
    const int ntile_n = p.Nc / WS_BN, nchunk = p.Ck / 16, ngrp = (nchunk + WS_CPW - 1) / WS_CPW, nmb = (HW + MB - 1) / MB;
    int bid = blockIdx.x;
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int cg = bid % ngrp; bid /= ngrp;
    const int mb = bid % nmb; const int n = bid / nmb;
    const int m0 = mb * MB, n0 = n_t * WS_BN;
    const int mlast = min(m0 + MB, HW) - 1;
    const int ylo = m0 / p.W - 1, yhi = mlast / p.W + 1;
    const int HP = p.W + 2, SLOTS = (yhi - ylo + 1) * HP;
    const float rHP = 1.f / (float)HP, rW = 1.f / (float)p.W;
    const int chunk = cg * WS_CPW + wave;
    const bool wave_live = chunk < nchunk;                             // (a contraction of 16 .. 48 channels: the last group is short)

    // ---- B: all nine taps of this wave's chunk, both pieces, issued before anything else ---------------------------------------------------------
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    const unsigned b_lane = (unsigned)(((lane >> 5) * p.Nc + n0 + (lane & 31)) * 16);
    const int b_chunk = 4 * p.Nc * 16, b_piece = 2 * p.Nc * 16;
    u32x4 breg[9][NB];
    if (wave_live) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < NB; ++e)
                breg[t][e] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_lane, (p.wtap[t] * nchunk + chunk) * b_chunk + e * b_piece, 0);
    }
    // ---- A: the halos of the group's chunks.  thread = (pixel of the pass, chunk c, k-octet): its eight channels and their modulation are fixed -----
    const int oct = tid & 1, ac = (tid >> 1) & (WS_CPW - 1), ps = tid >> 3;
    const int achunk = min(cg * WS_CPW + ac, nchunk - 1);
    const bool a_live = cg * WS_CPW + ac < nchunk;
    const float* xn = p.x + (int64_t)n * HW * p.ldx + achunk * 16 + oct * 8;
    float sv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sv[q] = 1.f;
    float smax = 1.f;
    if (p.in_scale != nullptr) {
        const float* sr = p.in_scale + (int64_t)n * p.Ck + achunk * 16 + oct * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(sr), s1 = *reinterpret_cast<const float4*>(sr + 4);
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
        float m = 0.f;                       // max|in_scale| over [N, Ck]: every workgroup derives the same value (split_act_kernel's recipe)
        for (int i = tid; i < p.N * p.Ck; i += 256) m = fmaxf(m, fabsf(p.in_scale[i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    const float mul = range_mul(*p.x_amax * p.x_amax_mul * smax);
    const float out_mul = 1.f / (mul * *p.w_scale);
    const int cplane = 4 * SLOTS * 16;                                 // bytes of one chunk's four (piece, k-octet) planes
    for (int s0 = 0; s0 < SLOTS; s0 += WS_SPP * WS_BATCH) {
        float4 raw[WS_BATCH][2];
#pragma unroll
        for (int k = 0; k < WS_BATCH; ++k) {
            const int slot = s0 + k * WS_SPP + ps;
            const int hy = ws_div(slot, HP, rHP), hx = slot - hy * HP;
            const int y = ylo + hy, x = hx - 1;
            raw[k][0] = make_float4(0.f, 0.f, 0.f, 0.f); raw[k][1] = raw[k][0];
            if (a_live && slot < SLOTS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                const float* src = xn + (int64_t)(y * p.W + x) * p.ldx;
                raw[k][0] = *reinterpret_cast<const float4*>(src); raw[k][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
#pragma unroll
        for (int k = 0; k < WS_BATCH; ++k) {
            const int slot = s0 + k * WS_SPP + ps;
            if (slot >= SLOTS) continue;
            float v[8] = {raw[k][0].x * sv[0], raw[k][0].y * sv[1], raw[k][0].z * sv[2], raw[k][0].w * sv[3],
                          raw[k][1].x * sv[4], raw[k][1].y * sv[5], raw[k][1].z * sv[6], raw[k][1].w * sv[7]};
            f16x8 h, l;
            split8(v, mul, h, l, 2048.f);
            *reinterpret_cast<f16x8*>(smem + ac * cplane + (oct * SLOTS + slot) * 16) = h;
            if constexpr (FULL) *reinterpret_cast<f16x8*>(smem + ac * cplane + ((2 + oct) * SLOTS + slot) * 16) = l;
        }
    }
    __syncthreads();

    // ---- 9 taps x MT x 3 MFMAs -----------------------------------------------------------------------------------------------------------------
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (wave_live) {
        unsigned a_addr[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = min(m0 + i * 32 + (lane & 31), HW - 1);
            const int y = ws_div(m, p.W, rW), x = m - y * p.W;
            a_addr[i] = (unsigned)(wave * cplane + ((lane >> 5) * SLOTS + (y - ylo) * HP + x + 1) * 16);
        }
        const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
        const int lo_plane = 2 * SLOTS * 16;
        // A fragments one tap ahead (one wave per SIMD: nothing else covers the LDS latency)
        f16x8 af[2][MT][NB];
        auto load_A = [&](int par, int t) {
            const int toff = (p.dy[t] * HP + p.dx[t]) * 16;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                af[par][i][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[i] + toff);
                if constexpr (FULL) af[par][i][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[i] + toff + lo_plane);
            }
        };
        load_A(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_A((t + 1) & 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 bh = ws_f16x8(breg[t][0]);
            f16x8 bl, bg;
            if constexpr (FULL) {
                bl = ws_f16x8(breg[t][1]);
                const f16x2* s2 = reinterpret_cast<const f16x2*>(&bh);
                f16x2* d2 = reinterpret_cast<f16x2*>(&bg);
#pragma unroll
                for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
            }
            if constexpr (FULL) {               // product-major: consecutive MFMAs write different accumulators
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][1], bg, acc[i], 0, 0, 0);       // small terms first
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][0], bl, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][0], bh, acc[i], 0, 0, 0);
        }
    }
    // ---- the four chunk tiles meet in LDS in wave order (fixed order: the sum a workgroup adds to `out` is a function of the geometry only), as an
    //      image of the accumulator registers ([tile][register quad][lane] float4: 16-byte conflict-free accesses); then every wave adds a quarter of
    //      the total to the pre-zeroed output: 32 lanes = 32 consecutive channels of one cell per atomic instruction ----------------------------------------------
    float4* img = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int w = 0; w < WS_CPW; ++w) {
        __syncthreads();                        // (w = 0: every wave is done reading the halos)
        if (wave != w) continue;
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) img[(i * 4 + g) * 64 + lane] = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = img[(i * 4 + g) * 64 + lane];
                    acc[i][4 * g] += o.x; acc[i][4 * g + 1] += o.y; acc[i][4 * g + 2] += o.z; acc[i][4 * g + 3] += o.w;
                    img[(i * 4 + g) * 64 + lane] = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                }
        }
    }
    __syncthreads();
    // every wave adds its share of the total (row tiles wave, wave + 4, ...; with fewer than four tiles, register quads) to `out`
    float* on = p.out + (int64_t)n * HW * p.ldo + n0 + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (((i * 4 + g) & 3) != wave) continue;            // 4 MT quads over 4 waves
            const float4 v = img[(i * 4 + g) * 64 + lane];
            const float vv[4] = {v.x, v.y, v.z, v.w};
            const int mrow = m0 + i * 32 + 4 * (lane >> 5) + 8 * g;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (mrow + q < HW) eg3d_acc(on + (int64_t)(mrow + q) * p.ldo, vv[q] * out_mul);
        }
}

int ws_lds_bytes(const eg3d_conv_ws_params& p, int MT) {
    const int HW = p.H * p.W, MB = 32 * MT;
    const int rows = (HW > MB ? MB / p.W : p.H) + 2;
    const int halo = WS_CPW * 4 * rows * (p.W + 2) * 16;
    return std::max(halo, MT * 4 * 64 * 16);
}

std::atomic<uint64_t> g_ws_attr[22];

template <bool FULL, int MT>
int launch_ws(const eg3d_conv_ws_params& p, hipStream_t st, int slot) {
    const int nmb = (p.H * p.W + 32 * MT - 1) / (32 * MT);
    const int blocks = p.N * nmb * ((p.Ck / 16 + WS_CPW - 1) / WS_CPW) * (p.Nc / WS_BN);
    const int lds = ws_lds_bytes(p, MT);
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_ws_kernel<FULL, MT>), lds, g_ws_attr[slot])) return e;
    hipLaunchKernelGGL((conv_ws_kernel<FULL, MT>), dim3(blocks), dim3(256), lds, st, p);
    return EG3D_OK;
}

// row tiles (32 cells) per workgroup: the smallest of 1 / 2 / 4 / 8 that holds the image, else 256-cell blocks of whole rows
int ws_mt(const eg3d_conv_ws_params& p) {
    const int HW = p.H * p.W;
    return HW <= 32 ? 1 : (HW <= 64 ? 2 : (HW <= 128 ? 4 : 8));
}


// ---- stride-2 adjoint: the data gradient of the up layers of the 8^2 .. 32^2 blocks (4^2 .. 16^2 output cells), same recipe ----------------------
//     out[n, a, b, o] += sum_t sum_k  g[n, 2a + dy[t], 2b + dx[t], k] * W[o, wtap[t], k]         dy, dx in {0, 1, 2}
// g: the FIR-adjointed gradient, fp32 NHWC [N, Hx, Wx, ldx] (Hx >= 2H + 1 rows are read where they exist, zeros elsewhere).  The halo of a chunk is
// the WHOLE image as its four parity images P[py][px][a'][b'] = g[2a' + py, 2b' + px] of (H + 1) x (W + 1) cells -- tap (dy, dx) then reads
// parity (dy & 1, dx & 1) at cell offset (dy >> 1, dx >> 1), a constant LDS offset, and a 32-cell MFMA row reads consecutive slots.  Four times
// the halo of the stride-1 form: CPW = 2 chunks per workgroup where four do not fit in LDS (16^2: 148 KB), and the 4 / CPW wave groups split the
// cells instead (MT tiles each).
template <bool FULL, int MT, int CPW>
__global__ void __launch_bounds__(256) conv_ws_s2adj_kernel(const eg3d_conv_ws_params p) {
    constexpr int NB = FULL ? 2 : 1;
    constexpr int NG = 4 / CPW;                                       // wave groups along the cells
    constexpr int SPP = 256 / (2 * CPW), BATCH = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = wave % CPW, mh = wave / CPW;
    const int HW = p.H * p.W, PW = p.W + 1, PP = (p.H + 1) * PW, SLOTS = 4 * PP;
    const float rPP = 1.f / (float)PP, rPW = 1.f / (float)PW, rW = 1.f / (float)p.W;
    const int ntile_n = p.Nc / WS_BN, nchunk = p.Ck / 16, ngrp = (nchunk + CPW - 1) / CPW;
    int bid = blockIdx.x;
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int cg = bid % ngrp; const int n = bid / ngrp;
    const int n0 = n_t * WS_BN;
    const int chunk = cg * CPW + cl;
    const bool wave_live = chunk < nchunk && mh * MT * 32 < HW;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    const unsigned b_lane = (unsigned)(((lane >> 5) * p.Nc + n0 + (lane & 31)) * 16);
    const int b_chunk = 4 * p.Nc * 16, b_piece = 2 * p.Nc * 16;
    u32x4 breg[9][NB];
    if (wave_live) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < NB; ++e)
                breg[t][e] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_lane, (p.wtap[t] * nchunk + chunk) * b_chunk + e * b_piece, 0);
    }
    const int oct = tid & 1, ac = (tid >> 1) & (CPW - 1), ps = tid / (2 * CPW);
    const int achunk = min(cg * CPW + ac, nchunk - 1);
    const bool a_live = cg * CPW + ac < nchunk;
    const float* xn = p.x + (int64_t)n * p.Hx * p.Wx * p.ldx + achunk * 16 + oct * 8;
    float sv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sv[q] = 1.f;
    float smax = 1.f;
    if (p.in_scale != nullptr) {
        const float* sr = p.in_scale + (int64_t)n * p.Ck + achunk * 16 + oct * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(sr), s1 = *reinterpret_cast<const float4*>(sr + 4);
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
        float m = 0.f;
        for (int i = tid; i < p.N * p.Ck; i += 256) m = fmaxf(m, fabsf(p.in_scale[i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    const float mul = range_mul(*p.x_amax * p.x_amax_mul * smax);
    const float out_mul = 1.f / (mul * *p.w_scale);
    const int cplane = 4 * SLOTS * 16;
    for (int s0 = 0; s0 < SLOTS; s0 += SPP * BATCH) {
        float4 raw[BATCH][2];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int slot = s0 + k * SPP + ps;
            const int par = ws_div(slot, PP, rPP), r = slot - par * PP;
            const int ap = ws_div(r, PW, rPW), bp = r - ap * PW;
            const int y = 2 * ap + (par >> 1), x = 2 * bp + (par & 1);
            raw[k][0] = make_float4(0.f, 0.f, 0.f, 0.f); raw[k][1] = raw[k][0];
            if (a_live && slot < SLOTS && y < p.Hx && x < p.Wx) {
                const float* src = xn + (int64_t)(y * p.Wx + x) * p.ldx;
                raw[k][0] = *reinterpret_cast<const float4*>(src); raw[k][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int slot = s0 + k * SPP + ps;
            if (slot >= SLOTS) continue;
            float v[8] = {raw[k][0].x * sv[0], raw[k][0].y * sv[1], raw[k][0].z * sv[2], raw[k][0].w * sv[3],
                          raw[k][1].x * sv[4], raw[k][1].y * sv[5], raw[k][1].z * sv[6], raw[k][1].w * sv[7]};
            f16x8 h, l;
            split8(v, mul, h, l, 2048.f);
            *reinterpret_cast<f16x8*>(smem + ac * cplane + (oct * SLOTS + slot) * 16) = h;
            if constexpr (FULL) *reinterpret_cast<f16x8*>(smem + ac * cplane + ((2 + oct) * SLOTS + slot) * 16) = l;
        }
    }
    __syncthreads();

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (wave_live) {
        unsigned a_addr[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = min((mh * MT + i) * 32 + (lane & 31), HW - 1);
            const int a = ws_div(m, p.W, rW), b = m - a * p.W;
            a_addr[i] = (unsigned)(cl * cplane + ((lane >> 5) * SLOTS + a * PW + b) * 16);
        }
        const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
        const int lo_plane = 2 * SLOTS * 16;
        f16x8 af[2][MT][NB];
        auto load_A = [&](int par, int t) {
            const int dy = p.dy[t], dx = p.dx[t];
            const int toff = (((dy & 1) * 2 + (dx & 1)) * PP + (dy >> 1) * PW + (dx >> 1)) * 16;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                af[par][i][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[i] + toff);
                if constexpr (FULL) af[par][i][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[i] + toff + lo_plane);
            }
        };
        load_A(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_A((t + 1) & 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 bh = ws_f16x8(breg[t][0]);
            f16x8 bl, bg;
            if constexpr (FULL) {
                bl = ws_f16x8(breg[t][1]);
                const f16x2* s2 = reinterpret_cast<const f16x2*>(&bh);
                f16x2* d2 = reinterpret_cast<f16x2*>(&bg);
#pragma unroll
                for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
            }
            if constexpr (FULL) {
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][1], bg, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][0], bl, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][0], bh, acc[i], 0, 0, 0);
        }
    }
    // the CPW chunk tiles of a wave group meet in LDS in chunk order (register images, one region per group), then every wave adds a quarter of
    // the quads to `out`
    float4* img = reinterpret_cast<float4*>(smem) + mh * MT * 4 * 64;
#pragma unroll
    for (int w = 0; w < CPW; ++w) {
        __syncthreads();
        if (cl != w) continue;
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) img[(i * 4 + g) * 64 + lane] = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = img[(i * 4 + g) * 64 + lane];
                    acc[i][4 * g] += o.x; acc[i][4 * g + 1] += o.y; acc[i][4 * g + 2] += o.z; acc[i][4 * g + 3] += o.w;
                    img[(i * 4 + g) * 64 + lane] = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
                }
        }
    }
    __syncthreads();
    const float4* all = reinterpret_cast<const float4*>(smem);
    float* on = p.out + (int64_t)n * HW * p.ldo + n0 + (lane & 31);
#pragma unroll
    for (int qd = 0; qd < NG * MT * 4; ++qd) {
        if ((qd & 3) != wave) continue;
        const float4 v = all[qd * 64 + lane];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const int mrow = (qd >> 2) * 32 + 4 * (lane >> 5) + 8 * (qd & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (mrow + q < HW) eg3d_acc(on + (int64_t)(mrow + q) * p.ldo, vv[q] * out_mul);
    }
}

// (MT, CPW) of the stride-2 adjoint form for an H x W output, 0 when it does not fit
void ws_s2_plan(const eg3d_conv_ws_params& p, int& mt, int& cpw, int& lds) {
    const int HW = p.H * p.W;
    mt = cpw = lds = 0;
    if (HW <= 32) { mt = 1; cpw = 4; } else if (HW <= 64) { mt = 2; cpw = 4; } else if (HW <= 128) { mt = 2; cpw = 2; } else if (HW <= 256) { mt = 4; cpw = 2; } else return;
    const int slots = 4 * (p.H + 1) * (p.W + 1);
    lds = std::max(cpw * 4 * slots * 16, (4 / cpw) * mt * 4 * 64 * 16);
}

template <bool FULL, int MT, int CPW>
int launch_ws_s2(const eg3d_conv_ws_params& p, hipStream_t st, int lds, int slot) {
    const int blocks = p.N * ((p.Ck / 16 + CPW - 1) / CPW) * (p.Nc / WS_BN);
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_ws_s2adj_kernel<FULL, MT, CPW>), lds, g_ws_attr[slot])) return e;
    hipLaunchKernelGGL((conv_ws_s2adj_kernel<FULL, MT, CPW>), dim3(blocks), dim3(256), lds, st, p);
    return EG3D_OK;
}


// ---- stride-2 TRANSPOSED conv (forward of the up layers of the 8^2 / 16^2 blocks: 4^2 / 8^2 input cells), same recipe ---------------------------------
//     out[n, 2a + ky, 2b + kx, o] += sum_k x[n, a, b, k] * in_scale[n, k] * W[o, wtap[3 ky + kx], k]          out: [N, 2H + 1, 2W + 1, ldo], pre-zeroed
// Output-centric like conv_v2_up.hip: the output pixels of parity (py, px) are the cells (a', b') of an (H + 1) x (W + 1) grid, cell (a', b') sums
// the taps with ky & 1 == py, kx & 1 == px at input (a' - (ky >> 1), b' - (kx >> 1)) -- four accumulator sets per wave (4 / 2 / 2 / 1 taps),
// one halo ((H + 2) x (W + 2) pixels: the one-pixel border supplies the zeros), nine weight fragments.  MT = ceil((H + 1)(W + 1) / 32) <= 3:
// the 16^2 -> 32^2 layer would need ten tiles per set and 4.5 M output atomics (19 us): it stays on the split-K implicit GEMM.
template <bool FULL, int MT>
__global__ void __launch_bounds__(256) conv_ws_up_kernel(const eg3d_conv_ws_params p) {
    constexpr int NB = FULL ? 2 : 1;
    constexpr int BATCH = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = p.H * p.W, PW = p.W + 1, PC = (p.H + 1) * PW;        // cells of a parity class
    const int HP = p.W + 2, SLOTS = (p.H + 2) * HP;
    const float rHP = 1.f / (float)HP, rPW = 1.f / (float)PW;
    const int Ho = 2 * p.H + 1, Wo = 2 * p.W + 1;
    const int ntile_n = p.Nc / WS_BN, nchunk = p.Ck / 16, ngrp = (nchunk + WS_CPW - 1) / WS_CPW;
    int bid = blockIdx.x;
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int cg = bid % ngrp; const int n = bid / ngrp;
    const int n0 = n_t * WS_BN;
    const int chunk = cg * WS_CPW + wave;
    const bool wave_live = chunk < nchunk;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    const unsigned b_lane = (unsigned)(((lane >> 5) * p.Nc + n0 + (lane & 31)) * 16);
    const int b_chunk = 4 * p.Nc * 16, b_piece = 2 * p.Nc * 16;
    u32x4 breg[9][NB];
    if (wave_live) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < NB; ++e)
                breg[t][e] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_lane, (p.wtap[t] * nchunk + chunk) * b_chunk + e * b_piece, 0);
    }
    const int oct = tid & 1, ac = (tid >> 1) & (WS_CPW - 1), ps = tid >> 3;
    const int achunk = min(cg * WS_CPW + ac, nchunk - 1);
    const bool a_live = cg * WS_CPW + ac < nchunk;
    const float* xn = p.x + (int64_t)n * HW * p.ldx + achunk * 16 + oct * 8;
    float sv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sv[q] = 1.f;
    float smax = 1.f;
    if (p.in_scale != nullptr) {
        const float* sr = p.in_scale + (int64_t)n * p.Ck + achunk * 16 + oct * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(sr), s1 = *reinterpret_cast<const float4*>(sr + 4);
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
        float m = 0.f;
        for (int i = tid; i < p.N * p.Ck; i += 256) m = fmaxf(m, fabsf(p.in_scale[i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    const float mul = range_mul(*p.x_amax * p.x_amax_mul * smax);
    const float out_mul = 1.f / (mul * *p.w_scale);
    const int cplane = 4 * SLOTS * 16;
    for (int s0 = 0; s0 < SLOTS; s0 += WS_SPP * BATCH) {
        float4 raw[BATCH][2];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int slot = s0 + k * WS_SPP + ps;
            const int hy = ws_div(slot, HP, rHP), hx = slot - hy * HP;
            const int y = hy - 1, x = hx - 1;
            raw[k][0] = make_float4(0.f, 0.f, 0.f, 0.f); raw[k][1] = raw[k][0];
            if (a_live && slot < SLOTS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                const float* src = xn + (int64_t)(y * p.W + x) * p.ldx;
                raw[k][0] = *reinterpret_cast<const float4*>(src); raw[k][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int slot = s0 + k * WS_SPP + ps;
            if (slot >= SLOTS) continue;
            float v[8] = {raw[k][0].x * sv[0], raw[k][0].y * sv[1], raw[k][0].z * sv[2], raw[k][0].w * sv[3],
                          raw[k][1].x * sv[4], raw[k][1].y * sv[5], raw[k][1].z * sv[6], raw[k][1].w * sv[7]};
            f16x8 h, l;
            split8(v, mul, h, l, 2048.f);
            *reinterpret_cast<f16x8*>(smem + ac * cplane + (oct * SLOTS + slot) * 16) = h;
            if constexpr (FULL) *reinterpret_cast<f16x8*>(smem + ac * cplane + ((2 + oct) * SLOTS + slot) * 16) = l;
        }
    }
    __syncthreads();

    f32x16 acc[4][MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;
    if (wave_live) {
        unsigned a_addr[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = min(i * 32 + (lane & 31), PC - 1);
            const int a = ws_div(m, PW, rPW), b = m - a * PW;
            a_addr[i] = (unsigned)(wave * cplane + ((lane >> 5) * SLOTS + (a + 1) * HP + b + 1) * 16);
        }
        const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
        const int lo_plane = 2 * SLOTS * 16;
        f16x8 af[2][MT][NB];
        auto load_A = [&](int par, int t) {
            const int toff = -(((t / 3) >> 1) * HP + ((t % 3) >> 1)) * 16;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                af[par][i][0] = *reinterpret_cast<const f16x8*>(smem + a_addr[i] + toff);
                if constexpr (FULL) af[par][i][1] = *reinterpret_cast<const f16x8*>(smem + a_addr[i] + toff + lo_plane);
            }
        };
        load_A(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            constexpr int dummy = 0; (void)dummy;
            if (t + 1 < 9) load_A((t + 1) & 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int P = ((t / 3) & 1) * 2 + ((t % 3) & 1);              // (compile-time in the unrolled loop)
            const f16x8 bh = ws_f16x8(breg[t][0]);
            f16x8 bl, bg;
            if constexpr (FULL) {
                bl = ws_f16x8(breg[t][1]);
                const f16x2* s2 = reinterpret_cast<const f16x2*>(&bh);
                f16x2* d2 = reinterpret_cast<f16x2*>(&bg);
#pragma unroll
                for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
            }
            if constexpr (FULL) {
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[P][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][1], bg, acc[P][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[P][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][0], bl, acc[P][i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[P][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t & 1][i][0], bh, acc[P][i], 0, 0, 0);
        }
    }
    // the four chunk tiles (4 parity sets x MT tiles each) meet in LDS in wave order; then every wave adds a quarter of the quads to `out`
    float4* img = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int w = 0; w < WS_CPW; ++w) {
        __syncthreads();
        if (wave != w) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4* q = img + ((c * MT + i) * 4 + g) * 64 + lane;
                    if (w > 0) {
                        const float4 o = *q;
                        acc[c][i][4 * g] += o.x; acc[c][i][4 * g + 1] += o.y; acc[c][i][4 * g + 2] += o.z; acc[c][i][4 * g + 3] += o.w;
                    }
                    *q = make_float4(acc[c][i][4 * g], acc[c][i][4 * g + 1], acc[c][i][4 * g + 2], acc[c][i][4 * g + 3]);
                }
    }
    __syncthreads();
    float* on = p.out + (int64_t)n * Ho * Wo * p.ldo + n0 + (lane & 31);
#pragma unroll
    for (int qd = 0; qd < 4 * MT * 4; ++qd) {
        if ((qd & 3) != wave) continue;
        const int tq = qd >> 2, c = tq / MT, i = tq - c * MT;
        const float4 v = img[qd * 64 + lane];
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const int m0r = i * 32 + 4 * (lane >> 5) + 8 * (qd & 3);
        int a = ws_div(m0r, PW, rPW), b = m0r - a * PW;                 // four consecutive cells: one division, then steps
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = 2 * a + (c >> 1), x = 2 * b + (c & 1);
            if (m0r + q < PC && y < Ho && x < Wo) eg3d_acc(on + (int64_t)(y * Wo + x) * p.ldo, vv[q] * out_mul);
            if (++b == PW) { b = 0; ++a; }
        }
    }
}

int ws_up_mt(const eg3d_conv_ws_params& p) { const int pc = (p.H + 1) * (p.W + 1); return pc <= 32 ? 1 : (pc <= 64 ? 2 : (pc <= 96 ? 3 : 0)); }
int ws_up_lds(const eg3d_conv_ws_params& p, int mt) { return std::max(WS_CPW * 4 * (p.H + 2) * (p.W + 2) * 16, 4 * mt * 4 * 64 * 16); }

template <bool FULL, int MT>
int launch_ws_up(const eg3d_conv_ws_params& p, hipStream_t st, int slot) {
    const int blocks = p.N * ((p.Ck / 16 + WS_CPW - 1) / WS_CPW) * (p.Nc / WS_BN);
    const int lds = ws_up_lds(p, MT);
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_ws_up_kernel<FULL, MT>), lds, g_ws_attr[slot])) return e;
    hipLaunchKernelGGL((conv_ws_up_kernel<FULL, MT>), dim3(blocks), dim3(256), lds, st, p);
    return EG3D_OK;
}

}  // namespace

extern "C" int eg3d_conv2d_ws_supported(const eg3d_conv_ws_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_ws_params& p = *pp;
    if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.W > 32 || p.Ck < 16 || (p.Ck & 15) || p.Nc < WS_BN || (p.Nc % WS_BN) || (p.ldx & 3) || p.ldx < p.Ck || p.ldo < p.Nc) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.out_stride == 2) {                                              // the transposed form: whole image per workgroup, <= 96 cells per output parity
        if (p.in_stride > 1) return 0;
        for (int t = 0; t < 9; ++t) if (p.wtap[t] < 0 || p.wtap[t] >= p.wtaps) return 0;
        const int mt = ws_up_mt(p);
        if (mt == 0 || ws_up_lds(p, mt) > 156 * 1024) return 0;
        if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
        return 1;
    }
    if (p.out_stride != 0 && p.out_stride != 1) return 0;
    if (p.in_stride == 2) {                                               // the stride-2 adjoint form: whole image per workgroup
        for (int t = 0; t < 9; ++t)
            if (p.dy[t] < 0 || p.dy[t] > 2 || p.dx[t] < 0 || p.dx[t] > 2 || p.wtap[t] < 0 || p.wtap[t] >= p.wtaps) return 0;
        if (p.Hx < 1 || p.Wx < 1 || (int64_t)p.N * p.Hx * p.Wx * p.ldx > 0x7fffffffll) return 0;
        int mt, cpw, lds;
        ws_s2_plan(p, mt, cpw, lds);
        if (mt == 0 || lds > 156 * 1024) return 0;
        if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
        return 1;
    }
    if (p.in_stride != 0 && p.in_stride != 1) return 0;
    for (int t = 0; t < 9; ++t)
        if (p.dy[t] < -1 || p.dy[t] > 1 || p.dx[t] < -1 || p.dx[t] > 1 || p.wtap[t] < 0 || p.wtap[t] >= p.wtaps) return 0;
    const int HW = p.H * p.W, MB = 32 * ws_mt(p);
    if (HW > MB && (MB % p.W)) return 0;                                  // blocks of whole rows
    const int rows = (HW > MB ? MB / p.W : p.H) + 2;
    if (rows * (p.W + 2) > WS_MAXSLOTS) return 0;
    if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.N * ((HW + MB - 1) / MB) * (p.Ck / 16) * (p.Nc / WS_BN) > 0x7fffffffll) return 0;
    if (ws_lds_bytes(p, ws_mt(p)) > 160 * 1024) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_ws(const eg3d_conv_ws_params* pp, void* stream) {
    if (!pp || !pp->x || !pp->w || !pp->out || !pp->x_amax || !pp->w_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_ws_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_ws_params& p = *pp;
    const void* ptrs[] = {p.x, p.in_scale, p.w};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    EG3D_DET_SCOPE(det, stream);
    EG3D_DET_BIND(det, p.out, p.out_stride == 2 ? (int64_t)p.N * (2 * p.H + 1) * (2 * p.W + 1) * p.ldo : (int64_t)p.N * p.H * p.W * p.ldo);
    EG3D_DET_COMMIT(det);
    const bool full = p.products != 1;
    int rc;
    if (p.out_stride == 2) {
        const int mt = ws_up_mt(p);
        if (mt == 1) rc = full ? launch_ws_up<true, 1>(p, st, 16) : launch_ws_up<false, 1>(p, st, 17);
        else if (mt == 2) rc = full ? launch_ws_up<true, 2>(p, st, 18) : launch_ws_up<false, 2>(p, st, 19);
        else rc = full ? launch_ws_up<true, 3>(p, st, 20) : launch_ws_up<false, 3>(p, st, 21);
        if (rc != EG3D_OK) return rc;
        EG3D_DET_END(det);
        return EG3D_OK;
    }
    if (p.in_stride == 2) {
        int mt, cpw, lds;
        ws_s2_plan(p, mt, cpw, lds);
        if (mt == 1) rc = full ? launch_ws_s2<true, 1, 4>(p, st, lds, 8) : launch_ws_s2<false, 1, 4>(p, st, lds, 9);
        else if (mt == 2 && cpw == 4) rc = full ? launch_ws_s2<true, 2, 4>(p, st, lds, 10) : launch_ws_s2<false, 2, 4>(p, st, lds, 11);
        else if (mt == 2) rc = full ? launch_ws_s2<true, 2, 2>(p, st, lds, 12) : launch_ws_s2<false, 2, 2>(p, st, lds, 13);
        else rc = full ? launch_ws_s2<true, 4, 2>(p, st, lds, 14) : launch_ws_s2<false, 4, 2>(p, st, lds, 15);
        if (rc != EG3D_OK) return rc;
        EG3D_DET_END(det);
        return EG3D_OK;
    }
    switch (ws_mt(p)) {
        case 1: rc = full ? launch_ws<true, 1>(p, st, 0) : launch_ws<false, 1>(p, st, 1); break;
        case 2: rc = full ? launch_ws<true, 2>(p, st, 2) : launch_ws<false, 2>(p, st, 3); break;
        case 4: rc = full ? launch_ws<true, 4>(p, st, 4) : launch_ws<false, 4>(p, st, 5); break;
        default: rc = full ? launch_ws<true, 8>(p, st, 6) : launch_ws<false, 8>(p, st, 7); break;
    }
    if (rc != EG3D_OK) return rc;
    EG3D_DET_END(det);
    return EG3D_OK;
}
