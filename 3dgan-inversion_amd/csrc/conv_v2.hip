// Pre-split, halo-staged implicit-GEMM convolution for gfx950 (the dominant kernel of the generator from round 2 on).
//
// Same arithmetic as the F16X3 mode of conv_igemm.hip -- every fp32 product is three v_mfma_f32_32x32x16_f16 products of two-piece
// fp16 operands, accumulated in fp32 -- but the operands arrive ALREADY split:
//   A  "split image" of the activation  [N][piece 2][Ck/8][Hi][Wi][8] fp16,  h = rtz16(x 2^ea),  l = rne16((x 2^ea - h) 2^11)
//   B  "split image" of the weights     [tap][Ck/16][piece 2][koct 2][Nc][8] fp16,  h = rtz16(w 2^eb),  l = rne16(w 2^eb - h)
//   product = h_a h_b + h_a l_b + l_a (h_b 2^-11)         (the scaled low piece keeps 22 significant bits down to |x 2^ea| = 2^-14)
// so the main loop contains no split arithmetic and no register -> LDS traffic at all:
//   * per 16-channel chunk the input HALO of the block's 8 x 32 output patch (<= 10 x 34 pixels) is brought into LDS once by LDS-DMA
//     (buffer_load ... lds; out-of-image pixels come back as zeros through the buffer bounds check) and is read by every tap with a
//     per-tap constant address offset -- an implicit GEMM re-reads (and re-splits) the same pixels once per tap;
//   * the 128-channel weight tile of one (tap, chunk) goes through a three-slot LDS ring, also by LDS-DMA;
//   * a step = one tap of one chunk = 24 MFMAs per wave (wave tile 128 cells x 64 channels), one s_barrier, counted s_waitcnt vmcnt.
// Block = 4 waves (2 x 2), tile 256 cells x 128 channels, 72 KB of LDS -> 2 blocks per CU.
// Measured on MI355X (tools/proto/conv_v2_proto.hip, 512^2 x 128 -> 128 and 256^2 x 256 -> 256): 375-400 TFLOP/s algorithmic =
// 1.13-1.2 PFLOP/s executed, against 1.5-1.6 PFLOP/s of a register-only MFMA loop on random data (the chip clocks down under dense
// fp16 MFMA load; 2.0-2.26 PFLOP/s on zeros) and 255-270 TFLOP/s of the loader-split kernel on the same layers.
//
// Replaces, like conv_igemm.hip, the F.conv2d / F.conv_transpose2d calls under modulated_conv2d (training/networks_stylegan2.py:34-91,
// torch_utils/ops/conv2d_resample.py:31-43,114-136) and their data gradient, for the layers whose grids fill the chip.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int PH = 8, PW = 32;                  // output patch of a block
constexpr int BN = 128;                         // output channels of a block
constexpr int A_PARTS = 6;                      // 64-slot wave-instructions per A plane (halo <= 10 x 34 = 340 <= 384 slots)
constexpr int APLANE = A_PARTS * 64 * 16;       // 6144 bytes
constexpr int BPLANE = BN * 16;                 // 2048
constexpr int ABUF = 4 * APLANE, BSLOT = 4 * BPLANE;
constexpr int LDS_A = 0, LDS_B = 2 * ABUF;
constexpr int LDS_MAIN = 2 * ABUF + 3 * BSLOT;  // 73728
constexpr int LDS_N = BN + 4;                   // epilogue staging row (floats)
constexpr int LDS_EPI = (3 * BN + 4 + 64 * LDS_N) * 4;  // column sums (ds, dbias, dd) + a scalar + 64 staged rows
constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, unsigned lds_byte, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 16, voff, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else static_assert(N == 0, "vmcnt immediate");
}

template <int NTAPS, bool FULL = true>         // FULL: three products per fp32 product; !FULL: high pieces only (EG3D_PREC_F16X1)
__global__ void __launch_bounds__(256, 2) conv_v2_kernel(const eg3d_conv_v2_params p, const int cls_base) {
    constexpr int APT = (6 + NTAPS - 1) / NTAPS;          // A parts a wave issues per step
    constexpr int NA_TAPS = 6 / APT;                      // ... during the first NA_TAPS taps of a chunk (APT divides 6)
    static_assert(6 % APT == 0, "A parts per step");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const eg3d_conv_class& cl = p.cls[cls_base + blockIdx.z];
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
    const int planeA = p.Hi * p.Wi * 16;                   // bytes of one (piece, k-octet) plane of the A image
    // tap extent of this class -> halo geometry
    int dymin = cl.dy[0], dymax = cl.dy[0], dxmin = cl.dx[0], dxmax = cl.dx[0];
#pragma unroll
    for (int t = 1; t < NTAPS; ++t) {
        dymin = min(dymin, cl.dy[t]); dymax = max(dymax, cl.dy[t]);
        dxmin = min(dxmin, cl.dx[t]); dxmax = max(dxmax, cl.dx[t]);
    }
    constexpr int hw = PW + 2;                                        // LDS row pitch of the halo: always 34 pixels (columns past the
    const int hh = PH + dymax - dymin;                                // tap extent are loaded but never read); <= 10 rows (host check)
    const int hslots = hw * hh;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;                  // LDS byte address of the dynamic array

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * planeA), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- A loader: wave w issues the wave-instructions j = w + 4 i (i = 0..5) of a chunk: plane j / 6, 64-slot part j % 6 ------------
    unsigned a_pix[6];
    int a_plane[6], a_part[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int j = wave + 4 * i;
        a_plane[i] = j / A_PARTS; a_part[i] = j % A_PARTS;
        const int slot = a_part[i] * 64 + lane;
        const int hy = slot / hw, hx = slot - hy * hw;
        const int y = y0 + dymin + hy, x = x0 + dxmin + hx;
        const bool ok = slot < hslots && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        a_pix[i] = ok ? (unsigned)((y * p.Wi + x) * 16) : OOB;
    }
    auto issue_A = [&](int chunk, int i) {
        const int piece = a_plane[i] >> 1, koct = a_plane[i] & 1;
        const unsigned plane_off = (unsigned)((((n * 2 + piece) * (p.Ck / 8)) + chunk * 2 + koct) * planeA);
        // (!FULL: the low-piece planes are never read -- the DMA slot is kept for the wait accounting but fetches nothing)
        glds16(ars, lds0 + LDS_A + (chunk & 1) * ABUF + a_plane[i] * APLANE + a_part[i] * 1024, (a_pix[i] == OOB || (!FULL && piece == 1)) ? OOB : a_pix[i] + plane_off);
    };
    auto issue_B = [&](int chunk, int tap, int slot) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = wave * 2 + e, plane = idx >> 1, half = idx & 1;
            const unsigned v = (unsigned)(((((cl.wtap[tap] * nchunk + chunk) * 4 + plane) * p.Nc) + n0 + half * 64 + lane) * 16);
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

    const unsigned a_lane = (unsigned)(((wm * 4 - dymin) * hw + (lane & 31) - dxmin) * 16 + (lane >> 5) * APLANE);
    const unsigned b_lane = (unsigned)((wn * 64 + (lane & 31)) * 16 + (lane >> 5) * BPLANE);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    // ---- prologue: A(0), B(step 0), B(step 1) --------------------------------------------------------------------------------------------
    const int S = nchunk * NTAPS;
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_A(0, i);
    issue_B(0, 0, 0);
    if (S > 1) issue_B(NTAPS > 1 ? 0 : 1, NTAPS > 1 ? 1 : 0, 1);
    else { issue_B(0, 0, 1); }                       // keeps the wait accounting uniform (never read)

    int step = 0;
    auto run_chunk = [&](const int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap, ++step) {
            // B(step) -- and at tap 0 all of A(chunk) -- were issued before B(step+1) [2 ops] and the A parts of the previous step
            if constexpr (NTAPS == 1) {
                wait_vm<0>();                        // one-tap classes (1/9 of an up-sampling layer): no look-ahead bookkeeping
            } else {
                if (LAST && tap == NTAPS - 1) wait_vm<0>();
                else if (!LAST && tap >= 1 && tap - 1 < NA_TAPS) wait_vm<2 + APT>();
                else wait_vm<2>();
            }
            __builtin_amdgcn_s_barrier();
            if (!LAST && tap < NA_TAPS) {
#pragma unroll
                for (int e = 0; e < APT; ++e) issue_A(chunk + 1, tap * APT + e);
            }
            {
                int t2 = tap + 2, c2 = chunk;
                while (t2 >= NTAPS) { t2 -= NTAPS; c2 += 1; }
                if (NTAPS == 1 ? c2 < nchunk : (!LAST || c2 == chunk)) issue_B(c2, t2, (step + 2) % 3);
            }
            const unsigned abase = LDS_A + (chunk & 1) * ABUF + a_lane + (unsigned)((cl.dy[tap] * hw + cl.dx[tap]) * 16);
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
        }
    };
    for (int chunk = 0; chunk + 1 < nchunk; ++chunk) run_chunk(chunk, std::false_type{});
    run_chunk(nchunk - 1, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: the tile goes through LDS once (64 rows at a time) so that every global access is 16 bytes per lane ----------------
    const float out_mul = 1.f / (*p.a_scale * *p.w_scale);              // exact powers of two
    const int epi = p.epi;
    float* ds_lds = reinterpret_cast<float*>(smem);
    float* db_lds = ds_lds + BN;
    float* dq_lds = ds_lds + 2 * BN;
    float* sc_lds = ds_lds + 3 * BN;
    float* stage = ds_lds + 3 * BN + 4;
    const bool act_on = epi == EG3D_EPI_BWD_ACT;                         // + the producing layer's activation backward (common.h)
    const bool bwd_like = epi == EG3D_EPI_BWD || act_on;
    const bool do_ds = bwd_like && p.ds != nullptr && p.xin != nullptr;
    const eg3d_act_bwd& ab = p.act_bwd;
    eg3d_act_bwd_consts abc = {};
    if (act_on) abc = eg3d_act_bwd_setup(ab);
    const bool row_sums = act_on && (ab.dnoise != nullptr || ab.dstrength != nullptr);
    if (tid < BN) { ds_lds[tid] = 0.f; db_lds[tid] = 0.f; dq_lds[tid] = 0.f; }
    if (tid == 0) sc_lds[0] = 0.f;
    const float strength = (epi == EG3D_EPI_FWD && p.noise != nullptr) ? *p.noise_strength : 0.f;
    const float act_slope = eg3d_act_pwl_slope(p.act, p.alpha);
    const int HWo = p.Ho * p.Wo;
    const int c4 = tid & 31;                            // this thread's float4 column group in every unit it handles
    const int col = n0 + c4 * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), scl4 = make_float4(1.f, 1.f, 1.f, 1.f), dsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (epi == EG3D_EPI_FWD && p.bias != nullptr) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
    if ((epi == EG3D_EPI_FWD || bwd_like) && p.out_scale != nullptr) scl4 = *reinterpret_cast<const float4*>(p.out_scale + (int64_t)n * p.Nc + col);
    float4 abd4 = make_float4(1.f, 1.f, 1.f, 1.f), abb4 = make_float4(0.f, 0.f, 0.f, 0.f), accb4 = abb4, accd4 = abb4;
    float accs = 0.f;
    if (act_on && ab.d != nullptr) abd4 = *reinterpret_cast<const float4*>(ab.d + (int64_t)n * p.Nc + col);
    if (act_on && ab.bias != nullptr) abb4 = *reinterpret_cast<const float4*>(ab.bias + col);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDS_N + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r] * out_mul;
        __syncthreads();
        // 64 rows x 32 float4 units = 2048 units, 8 per thread, in two groups of 4 (loads first, then arithmetic + stores)
#pragma unroll
        for (int ug = 0; ug < 8; ug += 4) {
            int offs[4], pixl[4];
            float4 va[4], sa[4], sb[4];
            float nz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = (tid + (ug + u) * 256) >> 5;             // 0..63: wave-row row >> 5, patch column row & 31
                const int ay = y0 + (row >> 5) * 4 + i, ax = x0 + (row & 31);
                const bool ok = ay < Ha && ax < Wa;
                const int pix = (n * p.Ho + ay * p.out_stride + cl.out_py) * p.Wo + ax * p.out_stride + cl.out_px;
                offs[u] = ok ? pix * p.ldo + col : -1;
                pixl[u] = pix - n * HWo;
                va[u] = *reinterpret_cast<const float4*>(stage + row * LDS_N + c4 * 4);
                sa[u] = make_float4(0.f, 0.f, 0.f, 0.f); sb[u] = sa[u]; nz[u] = 0.f;
                if (ok && (epi == EG3D_EPI_FWD || bwd_like) && p.addend != nullptr) sa[u] = *reinterpret_cast<const float4*>(p.addend + offs[u]);
                if (ok && epi == EG3D_EPI_FWD && p.noise != nullptr) nz[u] = p.noise[(int64_t)n * p.noise_nstride + pixl[u]];
                if (ok && act_on && ab.noise != nullptr) nz[u] = ab.noise[(int64_t)n * ab.noise_nstride + pixl[u]];
                if (ok && (do_ds || act_on)) sb[u] = *reinterpret_cast<const float4*>(p.xin + offs[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
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
                        if (row_sums) {             // the 32 lanes of a half-wave hold the 128 channels of this pixel (rows are half-wave uniform)
                            cs = eg3d_row_group_sum(cs, 32);
                            if (c4 == 0) {
                                if (ab.dnoise != nullptr) unsafeAtomicAdd(ab.dnoise + (int64_t)n * ab.dnoise_nstride + pixl[u], cs * abc.strength);
                                accs += cs * nz[u];
                            }
                        }
                    }
                }
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                *reinterpret_cast<float4*>(p.out + offs[u]) = v;
            }
        }
    }
    if (do_ds || act_on) {
        if (do_ds) {
            atomicAdd(&ds_lds[c4 * 4 + 0], dsum4.x); atomicAdd(&ds_lds[c4 * 4 + 1], dsum4.y);
            atomicAdd(&ds_lds[c4 * 4 + 2], dsum4.z); atomicAdd(&ds_lds[c4 * 4 + 3], dsum4.w);
        }
        if (act_on) {
            if (ab.dbias != nullptr) {
                atomicAdd(&db_lds[c4 * 4 + 0], accb4.x); atomicAdd(&db_lds[c4 * 4 + 1], accb4.y);
                atomicAdd(&db_lds[c4 * 4 + 2], accb4.z); atomicAdd(&db_lds[c4 * 4 + 3], accb4.w);
            }
            if (ab.dd != nullptr) {
                atomicAdd(&dq_lds[c4 * 4 + 0], accd4.x); atomicAdd(&dq_lds[c4 * 4 + 1], accd4.y);
                atomicAdd(&dq_lds[c4 * 4 + 2], accd4.z); atomicAdd(&dq_lds[c4 * 4 + 3], accd4.w);
            }
            if (ab.dstrength != nullptr && accs != 0.f) atomicAdd(sc_lds, accs);
        }
        __syncthreads();
        if (tid < BN) {
            if (do_ds) unsafeAtomicAdd(p.ds + (int64_t)n * p.Nc + n0 + tid, ds_lds[tid]);
            if (act_on && ab.dbias != nullptr) unsafeAtomicAdd(ab.dbias + n0 + tid, db_lds[tid]);
            if (act_on && ab.dd != nullptr)       // dL/dd = sum dy * z,  z = (pre - bias - noise) / d
                unsafeAtomicAdd(ab.dd + (int64_t)n * p.Nc + n0 + tid, dq_lds[tid] / (ab.d != nullptr ? ab.d[(int64_t)n * p.Nc + n0 + tid] : 1.f));
        }
        if (act_on && ab.dstrength != nullptr && tid == 0 && sc_lds[0] != 0.f) unsafeAtomicAdd(ab.dstrength, sc_lds[0]);
    }
    eg3d_commit_amax_block(amax, p.out_amax);    // max|out|: the consumer's operand range (one atomic per block)
}

// ---- operand preparation -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float* x, float mul, f16x8& h, f16x8& l, float lo_mul) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q] * mul, b = x[2 * q + 1] * mul;
        const fp16x2_t hh = __builtin_amdgcn_cvt_pkrtz(a, b);
        const float ra = __builtin_amdgcn_fmed3f((a - (float)hh[0]) * lo_mul, -65504.f, 65504.f);
        const float rb = __builtin_amdgcn_fmed3f((b - (float)hh[1]) * lo_mul, -65504.f, 65504.f);
        h[2 * q] = (_Float16)hh[0]; h[2 * q + 1] = (_Float16)hh[1];
        l[2 * q] = (_Float16)ra; l[2 * q + 1] = (_Float16)rb;
    }
}

// multiplier that brings `amax` to [2^13, 2^14): an exact power of two
__device__ __forceinline__ float range_mul(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);                   // amax = m 2^e, m in [0.5, 1)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, 14 - e);
}

// thread = pixel; loop over the channel octets of its NHWC row; writes are 16 bytes per lane, contiguous over the wave
__global__ void __launch_bounds__(256) split_act_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* x_amax, const float* s_amax,
                                                        f16x8* __restrict__ out, float* scale_out, int N, int HW, int C, int ldx) {
    float smax = 1.f;
    if (s_amax != nullptr) {
        smax = *s_amax;
    } else if (s != nullptr) {            // max|in_scale| over [N,C] (a few KB): every block derives the same value itself
        __shared__ float red[4];
        float m = 0.f;
        for (int i = threadIdx.x; i < N * C; i += 256) m = fmaxf(m, fabsf(s[i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    const float mul = range_mul(*x_amax * smax);
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = mul;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (int64_t)N * HW) return;
    const int n = (int)(pix / HW);
    const int64_t pp = pix - (int64_t)n * HW;
    const int noct = C / 8;
    const float* xr = x + pix * ldx;
    const float* sr = s ? s + (int64_t)n * C : nullptr;
    for (int ko = 0; ko < noct; ++ko) {
        float v[8];
        const float4 v0 = *reinterpret_cast<const float4*>(xr + ko * 8), v1 = *reinterpret_cast<const float4*>(xr + ko * 8 + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        if (sr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= sr[ko * 8 + q];
        }
        f16x8 h, l;
        split8(v, mul, h, l, 2048.f);
        out[((int64_t)(n * 2 + 0) * noct + ko) * HW + pp] = h;
        out[((int64_t)(n * 2 + 1) * noct + ko) * HW + pp] = l;
    }
}

__device__ __forceinline__ float finite_abs(float v) { const float a = fabsf(v); return a < 3.0e38f ? a : 0.f; }     // NaN / inf -> 0

__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n, float* out) {
    float m = 0.f;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = x4[i];
        m = fmaxf(m, fmaxf(fmaxf(finite_abs(v.x), finite_abs(v.y)), fmaxf(finite_abs(v.z), finite_abs(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, finite_abs(x[(n4 << 2) + threadIdx.x]));
    // one atomic per block: with one per wave a 150 K-element weight tensor spent 28 us queueing on a single address
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f && m < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
    }
}

// w: packed [O][T][I] fp32 -> [T][I/16][piece][koct][O] x 8 fp16, unscaled low piece
__global__ void __launch_bounds__(256) split_w_kernel(const float* __restrict__ w, const float* w_amax, f16x8* __restrict__ out, float* scale_out, int O, int I, int T, int w_row) {
    const float mul = range_mul(*w_amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = mul;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int noct = I / 8;
    if (t >= (int64_t)O * T * noct) return;
    const int o = (int)(t % O);
    const int64_t r = t / O;
    const int ko = (int)(r % noct), tap = (int)(r / noct);
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = w[(int64_t)o * w_row + (int64_t)tap * I + ko * 8 + q];
    f16x8 h, l;
    split8(v, mul, h, l, 1.f);
    const int chunk = ko >> 1, koct = ko & 1;
    out[((((int64_t)tap * (I / 16) + chunk) * 2 + 0) * 2 + koct) * O + o] = h;
    out[((((int64_t)tap * (I / 16) + chunk) * 2 + 1) * 2 + koct) * O + o] = l;
}

std::atomic<uint64_t> g_attr[5];

template <int NTAPS, bool FULL = true>
int launch_v2(const eg3d_conv_v2_params& p, int cls_base, int ncls, int max_tiles, hipStream_t st, int slot) {
    auto kern = conv_v2_kernel<NTAPS, FULL>;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_BYTES, g_attr[slot])) return e;
    hipLaunchKernelGGL(kern, dim3(max_tiles, 1, ncls), dim3(256), LDS_BYTES, st, p, cls_base);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

}  // namespace

extern "C" int eg3d_conv2d_v2_supported(const eg3d_conv_v2_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_v2_params& p = *pp;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck < 16 || (p.Ck & 15) || p.Nc < BN || (p.Nc % BN) || (p.ldo & 3)) return 0;
    if (p.in_stride != 1 || p.out_stride < 1 || p.ncls < 1 || p.ncls > 4) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.products == 1)                                  // the single-product instantiation exists for the 3x3 classes
        for (int c = 0; c < p.ncls; ++c) if (p.cls[c].ntaps != 9) return 0;
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_FWD && p.epi != EG3D_EPI_BWD && p.epi != EG3D_EPI_BWD_ACT) return 0;
    if (p.epi == EG3D_EPI_FWD && !eg3d_act_is_pwl(p.act)) return 0;
    if (p.epi == EG3D_EPI_BWD_ACT) {
        const eg3d_act_bwd& ab = p.act_bwd;
        if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return 0;          // invertible piecewise-linear activations only
        if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return 0;
    }
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.ntaps != 9 && k.ntaps != 4 && k.ntaps != 2 && k.ntaps != 1) return 0;
        int ymin = k.dy[0], ymax = k.dy[0], xmin = k.dx[0], xmax = k.dx[0];
        for (int t = 1; t < k.ntaps; ++t) { ymin = std::min(ymin, k.dy[t]); ymax = std::max(ymax, k.dy[t]); xmin = std::min(xmin, k.dx[t]); xmax = std::max(xmax, k.dx[t]); }
        if (ymax - ymin > 2 || xmax - xmin > 2) return 0;
        for (int t = 0; t < k.ntaps; ++t) if (k.wtap[t] < 0 || k.wtap[t] >= p.wtaps) return 0;
    }
    if ((int64_t)p.N * 2 * (p.Ck / 8) * p.Hi * p.Wi * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.N * p.Ho * p.Wo * p.ldo > INT32_MAX) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_v2(const eg3d_conv_v2_params* pp, void* stream) {
    if (!pp || !pp->a || !pp->w || !pp->out || !pp->a_scale || !pp->w_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_v2_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_v2_params& p = *pp;
    if (p.epi == EG3D_EPI_BWD_ACT && !p.xin) return EG3D_ERR_INVALID;
    const void* ptrs[] = {p.out, p.addend, p.xin, p.out_scale, p.bias, p.act_bwd.d, p.act_bwd.bias};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.epi == EG3D_EPI_FWD && p.noise && !p.noise_strength) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    // classes with the same tap count go into one launch (consecutive classes only: 9 | 4,2,2,1 | 1 ...)
    int c = 0;
    while (c < p.ncls) {
        int e = c;
        int max_tiles = 0;
        while (e < p.ncls && p.cls[e].ntaps == p.cls[c].ntaps) {
            const int t = p.N * eg3d_cdiv(p.cls[e].Ha, PH) * eg3d_cdiv(p.cls[e].Wa, PW) * (p.Nc / BN);
            max_tiles = std::max(max_tiles, t);
            ++e;
        }
        int rc;
        switch (p.cls[c].ntaps) {
            case 9: rc = p.products == 1 ? launch_v2<9, false>(p, c, e - c, max_tiles, st, 4) : launch_v2<9>(p, c, e - c, max_tiles, st, 0); break;
            case 4: rc = launch_v2<4>(p, c, e - c, max_tiles, st, 1); break;
            case 2: rc = launch_v2<2>(p, c, e - c, max_tiles, st, 2); break;
            default: rc = launch_v2<1>(p, c, e - c, max_tiles, st, 3); break;
        }
        if (rc != EG3D_OK) return rc;
        c = e;
    }
    return EG3D_OK;
}

extern "C" int64_t eg3d_split_activation_bytes(int N, int H, int W, int C) { return (int64_t)N * H * W * C * 4; }

extern "C" int eg3d_split_activation(const float* x, const float* in_scale, const float* x_amax, const float* s_amax, void* image, float* scale_out,
                                     int N, int H, int W, int C, int ldx, void* stream) {
    if (!x || !x_amax || !image || !scale_out || N <= 0 || H <= 0 || W <= 0 || C < 8 || (C & 7) || (ldx & 3) || ldx < C) return EG3D_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(image) & 15)) return EG3D_ERR_UNSUPPORTED;
    const int64_t pix = (int64_t)N * H * W;
    hipLaunchKernelGGL(split_act_kernel, dim3((unsigned)((pix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, in_scale, x_amax, s_amax,
                       reinterpret_cast<f16x8*>(image), scale_out, N, H * W, C, ldx);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_absmax(const float* x, int64_t n, float* out, void* stream) {
    if (!x || !out || n <= 0) return EG3D_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(x) & 15) return EG3D_ERR_UNSUPPORTED;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(512, n / (256 * 4 * 4)));       // >= 16 elements per thread
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_split_weight(const float* w, const float* w_amax, void* image, float* scale_out, int O, int I, int T, int w_row, void* stream) {
    if (!w || !w_amax || !image || !scale_out || O <= 0 || I < 16 || (I & 15) || T <= 0 || w_row < T * I) return EG3D_ERR_INVALID;
    const int64_t tot = (int64_t)O * T * (I / 8);
    hipLaunchKernelGGL(split_w_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w_amax, reinterpret_cast<f16x8*>(image), scale_out, O, I, T, w_row);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
