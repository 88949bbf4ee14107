// Data gradient of the up-sampling layers' transposed convolution = a stride-2 3x3 correlation (torch_utils/ops/conv2d_resample.py:114-136
// backward; the dX contract of conv2d_gradfix.py:139-143 for F.conv_transpose2d) on pre-split operands:
//     dx[n, a, b, ci] = sum_{ky,kx} sum_co  g[n, 2a + ky, 2b + kx, co] * w[co, ci, ky, kx],      g = FIR^T(dz), (2 Hi + 1) x (2 Wi + 1)
// With g stored as four PARITY images  g_p[a', b'] = g[2a' + py, 2b' + px]  (written directly by the FIR-adjoint pass, eg3d_fir44_adjoint_split,
// in the split fp16 layout -- no extra operand pass), tap (ky, kx) reads parity (ky & 1, kx & 1) at offset (ky >> 1, kx >> 1): unit-stride
// rows, so the machinery of conv_v2.hip applies -- 8 x 32 output patch x 128 channels per workgroup, LDS-DMA staged halos, weight tiles
// through a three-slot ring, 24 MFMAs per wave and step, one barrier per step, counted vmcnt waits, the same fused epilogues (style
// gradient, activation backward of the producing layer).  The K loop runs chunk-major, parity-minor: per 16-channel chunk four sub-chunks
// (parities) of 4 / 2 / 2 / 1 taps, each with its own halo (9 x 33 cells of ONE parity image), double-buffered; the next sub-chunk's halo
// is issued across the current one's steps.  The stride-2 correlation reads four input pixels per output cell where a stride-1 3x3 reads
// one: 4 x the L2 -> LDS operand traffic per MFMA is inherent.
// The loader-split kernel ran this gradient at 210-225 TFLOP/s (184 us on 513^2 x 128 -> 256^2 x 256, strided gathers split in the loader).
#include "conv_v2_common.h"

namespace {

constexpr int S_PARTS = 5;                      // 64-slot wave-instructions per A plane: halo 9 x 34 = 306 <= 320 slots
// static schedule of the nine steps of a chunk: weight tap 3 ky + kx, sub-chunk (parity 2 py + px), first step of its sub-chunk,
// A parts of the NEXT sub-chunk issued by each wave during the step (5 per wave and sub-chunk = 20 wave-instructions / 4 waves)
__device__ constexpr int TAP9[9] = {0, 2, 6, 8, 1, 7, 3, 5, 4};
__device__ constexpr int SUB[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
__device__ constexpr int AISS[9] = {2, 2, 1, 0, 3, 2, 3, 2, 5};
__device__ constexpr int AFIRST[9] = {0, 2, 4, 0, 0, 3, 0, 3, 0};
// ops allowed in flight when step s starts: B(s + 1) [2 per wave] + the A parts issued during step s - 1 unless they belong to the
// sub-chunk that starts now
__device__ constexpr int WAITN[9] = {2, 4, 4, 3, 2, 5, 2, 5, 2};
// ... derived from the schedule (the table above is checked against it): the previous step issued its A parts, then B(s + 1); the A parts may stay in flight
// unless their sub-chunk starts with this step; in the last chunk step 7 issues no B (there is no step 9) and step 8 waits for everything
constexpr int s2_allow(int s, bool last) {
    if (last && s == 8) return 0;
    const int ps = s >= 1 ? s - 1 : 8;
    return 2 + (SUB[s] != SUB[ps] ? 0 : AISS[ps]);
}
constexpr bool s2_table_ok() {
    for (int s = 0; s < 9; ++s)
        if (s2_allow(s, false) != WAITN[s]) return false;
    int parts[4] = {0, 0, 0, 0};
    for (int s = 0; s < 9; ++s) {                       // the parts of sub-chunk q are issued during sub-chunk q - 1, in order, S_PARTS per wave
        if (AISS[s] > 0 && AFIRST[s] != parts[(SUB[s] + 1) & 3]) return false;
        parts[(SUB[s] + 1) & 3] += AISS[s];
    }
    return parts[0] == S_PARTS && parts[1] == S_PARTS && parts[2] == S_PARTS && parts[3] == S_PARTS;
}
static_assert(s2_table_ok(), "conv_v2_s2adj: wait table / A-part schedule");

template <bool FULL, bool ATOMIC>
__global__ void __launch_bounds__(256, 2) conv_v2_s2adj_kernel(const eg3d_conv_v2_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const eg3d_conv_class& cl = p.cls[0];
    const int Ha = cl.Ha, Wa = cl.Wa;
    const int tiles_x = (Wa + PW - 1) / PW, tiles_y = (Ha + PH - 1) / PH, ntile_n = p.Nc / BN;
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
    const int Hp = p.Hi, Wp = p.Wi;                                  // dimensions of one parity image
    const int planeP = Hp * Wp * 16;                                 // bytes of one (piece, k-octet, parity) plane
    constexpr int hw = PW + 2;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * 4 * planeP), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- A loader: wave w issues the wave-instructions j = w + 4 i (i = 0..4) of a sub-chunk: plane j / 5, 64-slot part j % 5 ---------------
    unsigned a_pix[5];
    int a_plane[5], a_part[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int j = wave + 4 * i;
        a_plane[i] = j / S_PARTS; a_part[i] = j % S_PARTS;
        const int slot = a_part[i] * 64 + lane;
        const int hy = slot / hw, hx = slot - hy * hw;
        const int y = y0 + hy, x = x0 + hx;
        const bool ok = hy <= PH && y < Hp && x < Wp;              // cells a parity image does not have hold zeros (eg3d_fir44_adjoint_split)
        a_pix[i] = ok ? (unsigned)((y * Wp + x) * 16) : OOB;
    }
    auto issue_A = [&](int chunk, int par, int buf, int i) {
        const int piece = a_plane[i] >> 1, koct = a_plane[i] & 1;
        const unsigned plane_off = (unsigned)((((((n * 2 + piece) * (p.Ck / 8)) + chunk * 2 + koct) * 4 + par)) * planeP);
        glds16(ars, lds0 + LDS_A + buf * ABUF + a_plane[i] * APLANE + a_part[i] * 1024, (a_pix[i] == OOB || (!FULL && piece == 1)) ? OOB : a_pix[i] + plane_off);
    };
    int wtap_r[9];                       // (in scalar registers: through `cl` they are re-fetched from the kernel-argument segment after every boundary)
#pragma unroll
    for (int t = 0; t < 9; ++t) wtap_r[t] = __builtin_amdgcn_readfirstlane(cl.wtap[t]);
    auto issue_B = [&](int chunk, int t9, int slot) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = wave * 2 + e, plane = idx >> 1, half = idx & 1;
            const unsigned v = (unsigned)(((((wtap_r[t9] * nchunk + chunk) * 4 + plane) * p.Nc) + n0 + half * 64 + lane) * 16);
            glds16(wrs, lds0 + LDS_B + slot * BSLOT + plane * BPLANE + half * 1024, (!FULL && plane >= 2) ? OOB : v);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned a_lane = (unsigned)(((wm * 4) * hw + (lane & 31)) * 16 + (lane >> 5) * APLANE);
    const unsigned b_lane = (unsigned)((wn * 64 + (lane & 31)) * 16 + (lane >> 5) * BPLANE);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    // ---- prologue: A(c0, parity 0), B(step 0), B(step 1) ---------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_A(c0, 0, 0, i);
    issue_B(c0, TAP9[0], 0);
    issue_B(c0, TAP9[1], 1);

    int step = 0;                        // global step counter: weight ring slot = step % 3; sub-chunk counter = 4 (chunk - c0) + SUB -> A buffer & 1
    auto run_chunk = [&](const int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        static_for<0, 9>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            step_sync<s2_allow(s, LAST)>();          // conv_v2_common.h: this step's tiles have landed, everybody's LDS reads of the previous step have returned
            const int sub = SUB[s];
            const int abufi = sub & 1;                       // four sub-chunks per chunk: the buffer parity restarts with every chunk
            // A parts of the next sub-chunk (the next chunk's parity 0 after the last one)
            if (!(LAST && sub == 3)) {
#pragma unroll
                for (int e = 0; e < AISS[s]; ++e) issue_A(sub == 3 ? chunk + 1 : chunk, sub == 3 ? 0 : sub + 1, abufi ^ 1, AFIRST[s] + e);
            }
            if constexpr (s + 2 < 9) issue_B(chunk, TAP9[s + 2], (step + 2) % 3);
            else if constexpr (!LAST) issue_B(chunk + 1, TAP9[s + 2 - 9], (step + 2) % 3);
            const int t9 = TAP9[s];
            const int oy = (t9 / 3) >> 1, ox = (t9 % 3) >> 1;
            const unsigned abase = LDS_A + abufi * ABUF + a_lane + (unsigned)((oy * hw + ox) * 16);
            const unsigned bbase = LDS_B + (step % 3) * BSLOT + b_lane;
            f16x8 bh[2], bl[2], bg[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(smem + bbase + j * 512);
                if constexpr (FULL) {
                    bl[j] = *reinterpret_cast<const f16x8*>(smem + bbase + j * 512 + 2 * BPLANE);
                    f16x2* s2 = reinterpret_cast<f16x2*>(&bh[j]);
                    f16x2* d2 = reinterpret_cast<f16x2*>(&bg[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + abase + i * hw * 16);
                if constexpr (FULL) {
                    const f16x8 al = *reinterpret_cast<const f16x8*>(smem + abase + i * hw * 16 + 2 * APLANE);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {       // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bg[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
            ++step;
        });
    };
    for (int chunk = c0; chunk + 1 < c1; ++chunk) run_chunk(chunk, std::false_type{});
    run_chunk(c1 - 1, std::true_type{});
    step_sync<0>();                      // every LDS-DMA and LDS read of the main loop is over: the epilogue re-uses the dynamic LDS
    v2_epilogue<ATOMIC, 4>(p, acc, Ha, Wa, cl.out_py, cl.out_px, n, y0, x0, n0, smem, 1.f / (*p.a_scale * *p.w_scale));
}

// ---- FIR adjoint of an up layer + parity split --------------------------------------------------------------------------------------------
// g = upfirdn2d(dz, outer([1,3,3,1]) / 64, padding 2, gain) on [N, 2 Hi, 2 Wi, C] -> (2 Hi + 1) x (2 Wi + 1), written as the four parity images
// in the split fp16 layout [N][piece][C/8][parity][Hp = Hi + 1][Wp = Wi + 1][8], range-normalised by the bound max|g| <= gain * max|dz|
// (a non-negative filter of unit sum).  g[2a + py, 2b + px] = gain * sum_{i,j<4} k[i] k[j] dz[2a + py - 2 + i, 2b + px - 2 + j].
// Block = FA cell rows x 32 cells x 64 channels.  Phase 1 (lanes along channels: whole 256-byte runs of a pixel): each thread walks the
// 2 FA + 3 input rows of its (column, channel quad) and leaves the vertical sums of both row parities in LDS.  Phase 2 (lanes along cells:
// 16-byte stores contiguous over a plane's row): horizontal sums, split, store.  The thread-per-(cell, octet) form without LDS measured
// ~5x slower (32-byte reads from 64 different pixel rows per instruction, 16-byte stores into 16 different planes).
#ifndef UE_XCD_REMAP
#define UE_XCD_REMAP 1
#endif
constexpr int FA = 2, FW = 32, FCH = 64, FCOLS = 2 * FW + 3;        // input columns 2 b0 - 2 .. 2 b0 + 2 FW
__global__ void __launch_bounds__(256) fir44_adjoint_split_kernel(const float* __restrict__ dz, const float* dz_amax, f16x8* __restrict__ out, float* scale_out,
                                                                  int N, int Hi, int Wi, int C, int ldz, float gain) {
    constexpr int PITCH = FCH + 4;
    __shared__ __attribute__((aligned(16))) float V[FA * 2 * FCOLS * PITCH];          // [cell row][py][column][channel]
    const float mul = range_mul(*dz_amax * gain);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *scale_out = mul;
    const int Hp = Hi + 1, Wp = Wi + 1, Ho = 2 * Hi, Wo = 2 * Wi;
    const int tiles_x = (Wp + FW - 1) / FW, tiles_y = (Hp + FA - 1) / FA;
    const int bid = UE_XCD_REMAP ? eg3d_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;       // vertical neighbours (3 shared input rows) on one XCD: see upconv_epilogue_kernel
    const int n = bid / (tiles_x * tiles_y);
    const int t = bid - n * tiles_x * tiles_y;
    const int a0 = (t / tiles_x) * FA, b0 = (t % tiles_x) * FW;
    const int c0 = blockIdx.y * FCH;
    const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    // ---- phase 1: vertical sums.  item = (column, channel quad): 67 x 16 items over 256 threads
    for (int it = threadIdx.x; it < FCOLS * (FCH / 4); it += 256) {
        const int col = it / (FCH / 4), c4 = it - col * (FCH / 4);
        const int x = 2 * b0 - 2 + col;
        float4 r[2 * FA + 3];
#pragma unroll
        for (int q = 0; q < 2 * FA + 3; ++q) {
            const int y = 2 * a0 - 2 + q;
            r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)y < (unsigned)Ho && (unsigned)x < (unsigned)Wo) r[q] = *reinterpret_cast<const float4*>(dz + ((int64_t)(n * Ho + y) * Wo + x) * ldz + c0 + c4 * 4);
        }
#pragma unroll
        for (int ar = 0; ar < FA; ++ar)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int q0 = 2 * ar + py;                         // rows 2 (a0 + ar) + py - 2 + i  =  index q0 + i
                float4 s;
                s.x = (r[q0].x * k[0] + r[q0 + 1].x * k[1]) + (r[q0 + 2].x * k[2] + r[q0 + 3].x * k[3]);
                s.y = (r[q0].y * k[0] + r[q0 + 1].y * k[1]) + (r[q0 + 2].y * k[2] + r[q0 + 3].y * k[3]);
                s.z = (r[q0].z * k[0] + r[q0 + 1].z * k[1]) + (r[q0 + 2].z * k[2] + r[q0 + 3].z * k[3]);
                s.w = (r[q0].w * k[0] + r[q0 + 1].w * k[1]) + (r[q0 + 2].w * k[2] + r[q0 + 3].w * k[3]);
                *reinterpret_cast<float4*>(V + ((ar * 2 + py) * FCOLS + col) * PITCH + c4 * 4) = s;
            }
    }
    __syncthreads();
    // ---- phase 2: horizontal sums + split.  item = (cell row, py, octet, cell): lanes along the 32 cells
    const float gm = gain * mul;
    const int noct = C / 8;
    for (int it = threadIdx.x; it < FA * 2 * (FCH / 8) * FW; it += 256) {
        const int bl = it % FW;
        int rest = it / FW;
        const int kl = rest % (FCH / 8); rest /= (FCH / 8);
        const int py = rest & 1, ar = rest >> 1;
        const int a = a0 + ar, b = b0 + bl;
        if (a >= Hp || b >= Wp) continue;
        const float* vp = V + ((ar * 2 + py) * FCOLS + 2 * bl) * PITCH + kl * 8;           // columns 2 bl .. 2 bl + 4  <->  x = 2 b - 2 .. 2 b + 2
        float v[5][8];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float4 lo = *reinterpret_cast<const float4*>(vp + j * PITCH), hi = *reinterpret_cast<const float4*>(vp + j * PITCH + 4);
            v[j][0] = lo.x; v[j][1] = lo.y; v[j][2] = lo.z; v[j][3] = lo.w; v[j][4] = hi.x; v[j][5] = hi.y; v[j][6] = hi.z; v[j][7] = hi.w;
        }
        const int ko = c0 / 8 + kl;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const bool exists = (a < Hi || py == 0) && (b < Wi || px == 0);
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float h = (v[px][q] * k[0] + v[px + 1][q] * k[1]) + (v[px + 2][q] * k[2] + v[px + 3][q] * k[3]);
                o[q] = exists ? h : 0.f;
            }
            f16x8 h, l;
            split8(o, gm, h, l, 2048.f);
            const int par = py * 2 + px;
            out[((((int64_t)(n * 2 + 0) * noct + ko) * 4 + par) * Hp + a) * Wp + b] = h;
            out[((((int64_t)(n * 2 + 1) * noct + ko) * 4 + par) * Hp + a) * Wp + b] = l;
        }
    }
}

std::atomic<uint64_t> g_attr_adj[4];

}  // namespace

extern "C" int eg3d_conv2d_v2_s2adj_supported(const eg3d_conv_v2_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_v2_params& p = *pp;
    if (p.N <= 0 || p.Hi < 2 || p.Wi < 2 || p.Ck < 16 || (p.Ck & 15) || p.Nc < BN || (p.Nc % BN) || (p.ldo & 3)) return 0;
    if (p.ncls != 1 || p.cls[0].ntaps != 9 || p.out_stride < 1 || p.wtaps < 9) return 0;
    const eg3d_conv_class& k = p.cls[0];
    for (int t = 0; t < 9; ++t) if (k.dy[t] != t / 3 || k.dx[t] != t % 3 || k.wtap[t] < 0 || k.wtap[t] >= p.wtaps) return 0;     // g[2a + ky, 2b + kx]
    if (k.Ha > p.Hi - 1 || k.Wa > p.Wi - 1 || k.Ha < 1 || k.Wa < 1) return 0;                      // parity images of Ha + 1 x Wa + 1 cells (or larger)
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_FWD && p.epi != EG3D_EPI_BWD && p.epi != EG3D_EPI_BWD_ACT && p.epi != EG3D_EPI_ATOMIC) return 0;
    if (p.ksplit > 1 && (p.epi != EG3D_EPI_ATOMIC || p.ksplit > p.Ck / 16 || p.ksplit > 65535)) return 0;
    if (p.epi == EG3D_EPI_FWD && !eg3d_act_is_pwl(p.act)) return 0;
    if (p.epi == EG3D_EPI_BWD_ACT) {
        const eg3d_act_bwd& ab = p.act_bwd;
        if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return 0;
        if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return 0;
    }
    if ((int64_t)p.N * 2 * (p.Ck / 8) * 4 * p.Hi * p.Wi * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.N * p.Ho * p.Wo * p.ldo > INT32_MAX) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_v2_s2adj(const eg3d_conv_v2_params* pp, void* stream) {
    if (!pp || !pp->a || !pp->w || !pp->out || !pp->a_scale || !pp->w_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_v2_s2adj_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_v2_params& p = *pp;
    if (p.epi == EG3D_EPI_BWD_ACT && !p.xin) return EG3D_ERR_INVALID;
    const void* ptrs[] = {p.out, p.addend, p.xin, p.out_scale, p.bias, p.act_bwd.d, p.act_bwd.bias};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    const int tiles = p.N * eg3d_cdiv(p.cls[0].Ha, PH) * eg3d_cdiv(p.cls[0].Wa, PW) * (p.Nc / BN);
    const dim3 grid(tiles, p.ksplit > 1 ? p.ksplit : 1, 1);
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&](auto kern, int slot) -> int {
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, g_attr_adj[slot])) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, st, p);
        return 0;
    };
    const bool at = p.epi == EG3D_EPI_ATOMIC;
    int rc;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND_V2(det, p); EG3D_DET_COMMIT(det);
    if (p.products == 1) rc = at ? launch(conv_v2_s2adj_kernel<false, true>, 3) : launch(conv_v2_s2adj_kernel<false, false>, 1);
    else rc = at ? launch(conv_v2_s2adj_kernel<true, true>, 2) : launch(conv_v2_s2adj_kernel<true, false>, 0);
    if (rc) return rc;
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int64_t eg3d_fir44_adjoint_split_bytes(int N, int Hi, int Wi, int C) { return (int64_t)N * 2 * (C / 8) * 4 * (Hi + 1) * (Wi + 1) * 16; }

extern "C" int eg3d_fir44_adjoint_split(const float* dz, const float* dz_amax, void* image, float* scale_out, int N, int Hi, int Wi, int C, int ldz, float gain,
                                        void* stream) {
    if (!dz || !dz_amax || !image || !scale_out || N <= 0 || Hi <= 0 || Wi <= 0 || C < 8 || (C & 7) || (ldz & 3) || ldz < C || !(gain > 0.f)) return EG3D_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(dz) & 15) || (reinterpret_cast<uintptr_t>(image) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (C % FCH) return EG3D_ERR_UNSUPPORTED;
    const int tiles = N * eg3d_cdiv(Hi + 1, FA) * eg3d_cdiv(Wi + 1, FW);
    hipLaunchKernelGGL(fir44_adjoint_split_kernel, dim3(tiles, C / FCH), dim3(256), 0, (hipStream_t)stream, dz, dz_amax,
                       reinterpret_cast<f16x8*>(image), scale_out, N, Hi, Wi, C, ldz, gain);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
