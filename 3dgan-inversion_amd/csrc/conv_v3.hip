// Wave-split pre-split convolution for gfx950: the 3x3 layers whose grids cannot fill the chip with the 256-cell x 128-channel tiles of
// conv_v2.hip (64^2 x 512, 32^2 x 512 and the 128^2 / 256^2 backbone layers at one image per GPU; training/networks_stylegan2.py:34-91,417-461).
//
// Same operands and arithmetic as conv_v2.hip (two-piece fp16 split images, three v_mfma_f32_32x32x16_f16 products per fp32 product, fp32
// accumulation, fused forward / data-gradient epilogues), different decomposition:
//   * workgroup tile = RPW x 32 cells x 64 channels (128 or 64 cells): 64^2 x 512 -> 512 gives 256 workgroups WITHOUT a cross-workgroup
//     split of the contraction -- no zero fill, no atomics, no slabs / tickets, no finishing pass, fused epilogue intact;
//   * the contraction is split over the NW waves of the workgroup instead: wave w owns the 16-channel chunks c = w (mod NW) and computes the
//     WHOLE tile for them (24 or 12 MFMAs per (tap, chunk) step).  The waves share nothing in the main loop, so there is no barrier in it:
//       - A: each wave stages the halo of ITS chunks into a private double-buffered LDS area by LDS-DMA (the next chunk's halo is issued
//            two instructions per tap while the current one is multiplied) and waits on its own vmcnt only;
//       - B: nobody else needs this wave's weight tile, so it never touches LDS: the MFMA B fragment of a lane is 16 contiguous bytes of the
//            weight image ([tap][chunk][piece][koct][Nc][8]), 32 lanes = one 512-byte run -- buffer_load_dwordx4 straight into VGPRs,
//            two steps ahead;
//   * after the last step the NW partial tiles meet in LDS (each wave writes the tiles it does not own, the owner adds them in wave order:
//     the result is a fixed function of the geometry, run-to-run deterministic), the sum is staged as [cell][64] and the epilogue of
//     conv_v2_common.h's form (16-byte global accesses, loads first) runs on it with all waves.
//   * n_t (the 64-channel tile) is the fastest index of blockIdx.x: the hardware places block b on XCD b % 8, so every XCD works on one or
//     two channel tiles and its 1.2 MB weight slice stays in its own L2 while the activation image streams through.
// One wave per SIMD (104 KB of LDS per workgroup): the schedule inside the wave has to cover its own latencies -- the B prefetch distance
// and the per-tap DMA issue are what does that.
#include "conv_v2_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int BN3 = 64;                         // output channels of a workgroup
constexpr int HW3 = PW + 2;                     // halo row pitch in pixels
constexpr int STG_N = BN3 + 4;                  // staging row pitch (floats)

template <int RPW, bool FULL = true> struct v3g {
    static constexpr int SLOTS = (RPW + 2) * HW3;              // halo pixels of a chunk: 204 | 136
    static constexpr int APL = SLOTS * 16;                     // bytes of one (piece, k-octet) plane
    // the four planes of a chunk are ONE run of 4 SLOTS 16-byte slots, covered by 64-slot DMA instructions that may straddle planes (every lane
    // carries its own source offset); only the last instruction is partial, and its surplus lanes (source out of bounds: zeros) land in the pad
    // behind the buffer.  (Exec-masked partial instructions per plane were tail-merged by the compiler with their unconditional neighbours into
    // ONE instruction whose LDS base -- M0, uniform by construction -- became the first active lane's: wrong destination for the other lanes.)
    static constexpr int NOPS = (4 * SLOTS + 63) / 64;         // 13 | 9
    static constexpr int NOPS_USED = FULL ? NOPS : (2 * SLOTS + 63) / 64;      // single-product arithmetic never reads the low-piece planes
    static constexpr int ABUF3 = NOPS * 1024;                  // 13312 | 9216
    static constexpr int WAVE_LDS = 2 * ABUF3;                 // double-buffered
    static constexpr int STAGE = RPW * 32 * STG_N * 4;         // reduced tile as [cell][64 + 4] floats
};
template <int RPW, int NW> constexpr int v3_lds_bytes() {
    constexpr int main_b = NW * v3g<RPW>::WAVE_LDS;
    constexpr int red_b = 2 * RPW * (NW - 1) * 4096;           // partial tiles that travel: T (NW - 1) blocks of 16 x 64 floats
    constexpr int stage_b = v3g<RPW>::STAGE;
    constexpr int m1 = main_b > red_b ? main_b : red_b;
    return m1 > stage_b ? m1 : stage_b;
}

template <int N> __device__ __forceinline__ void wait_vm3() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 0, "vmcnt immediate");
}

__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
// LDS-DMA with a scalar offset: 16 bytes per lane from rs[voff + soff] to lds_byte + 16 lane
__device__ __forceinline__ void glds16s(__amdgpu_buffer_rsrc_t rs, unsigned lds_byte, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(uintptr_t)lds_byte, 16, voff, soff, 0, 0);
}

// ---- epilogue on the reduced tile stage[cell][STG_N]: forward / data-gradient / + producer's activation backward (conv_v2_common.h's, for a
//      64-channel tile: 16 lanes hold a pixel's channels, NT / 16 pixels per pass) ------------------------------------------------------
template <int RPW, int NW>
__device__ __forceinline__ void v3_epilogue(const eg3d_conv_v2_params& p, const float* stage, const int Ha, const int Wa, const int out_py, const int out_px,
                                            const int n, const int y0, const int x0, const int n0) {
    constexpr int NT = NW * 64, RPP = NT / 16, NU = RPW * 32 / RPP;     // threads, rows per pass, units per thread (8 | 4 | 2)
    constexpr int UG = NU < 4 ? NU : 4;
    __shared__ float ds_lds[BN3], db_lds[BN3], dq_lds[BN3], sc_lds[1], nzl[4 * 32];
    const int tid = threadIdx.x;
    const int epi = p.epi;
    const bool act_on = epi == EG3D_EPI_BWD_ACT;
    const bool bwd_like = epi == EG3D_EPI_BWD || act_on;
    const bool do_ds = bwd_like && p.ds != nullptr && p.xin != nullptr;
    const eg3d_act_bwd& ab = p.act_bwd;
    eg3d_act_bwd_consts abc = {};
    if (act_on) abc = eg3d_act_bwd_setup(ab);
    const bool row_sums = act_on && (ab.dnoise != nullptr || ab.dstrength != nullptr);
    if (tid < BN3) { ds_lds[tid] = 0.f; db_lds[tid] = 0.f; dq_lds[tid] = 0.f; }
    if (tid == 0) sc_lds[0] = 0.f;
    const float strength = (epi == EG3D_EPI_FWD && p.noise != nullptr) ? *p.noise_strength : 0.f;
    const float act_slope = eg3d_act_pwl_slope(p.act, p.alpha);
    const int HWo = p.Ho * p.Wo;
    const int c4 = tid & 15;
    const int col = n0 + c4 * 4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), scl4 = make_float4(1.f, 1.f, 1.f, 1.f), dsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (epi == EG3D_EPI_FWD && p.bias != nullptr) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
    if ((epi == EG3D_EPI_FWD || bwd_like) && p.out_scale != nullptr) scl4 = *reinterpret_cast<const float4*>(p.out_scale + (int64_t)n * p.Nc + col);
    float4 abd4 = make_float4(1.f, 1.f, 1.f, 1.f), abb4 = make_float4(0.f, 0.f, 0.f, 0.f), accb4 = abb4, accd4 = abb4;
    float accs = 0.f;
    if (act_on && ab.d != nullptr) abd4 = *reinterpret_cast<const float4*>(ab.d + (int64_t)n * p.Nc + col);
    if (act_on && ab.bias != nullptr) abb4 = *reinterpret_cast<const float4*>(ab.bias + col);
    float amax = 0.f;
    {
        const float* nsrc = (epi == EG3D_EPI_FWD && p.noise != nullptr) ? p.noise + (int64_t)n * p.noise_nstride
                          : ((act_on && ab.noise != nullptr) ? ab.noise + (int64_t)n * ab.noise_nstride : nullptr);
        if (tid < RPW * 32) {
            const int ay = y0 + (tid >> 5), ax = x0 + (tid & 31);
            float v = 0.f;
            if (nsrc != nullptr && ay < Ha && ax < Wa) v = nsrc[(ay * p.out_stride + out_py) * p.Wo + ax * p.out_stride + out_px];
            nzl[tid] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ug = 0; ug < NU; ug += UG) {
        int offs[UG], pixl[UG];
        float4 va[UG], sa[UG], sb[UG];
        float nz[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int row = (tid >> 4) + (ug + u) * RPP;             // patch cell 0 .. RPW * 32 - 1
            const int ay = y0 + (row >> 5), ax = x0 + (row & 31);
            const bool ok = ay < Ha && ax < Wa;
            const int pix = (n * p.Ho + ay * p.out_stride + out_py) * p.Wo + ax * p.out_stride + out_px;
            offs[u] = ok ? pix * p.ldo + col : -1;
            pixl[u] = pix - n * HWo;
            va[u] = *reinterpret_cast<const float4*>(stage + row * STG_N + c4 * 4);
            sa[u] = make_float4(0.f, 0.f, 0.f, 0.f); sb[u] = sa[u];
            if (ok && (epi == EG3D_EPI_FWD || bwd_like) && p.addend != nullptr) sa[u] = *reinterpret_cast<const float4*>(p.addend + offs[u]);
            nz[u] = nzl[row];
            if (ok && (do_ds || act_on)) sb[u] = *reinterpret_cast<const float4*>(p.xin + offs[u]);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
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
                if (act_on) {
                    float cs;
                    v = eg3d_act_bwd_unit(abc, v, sb[u], abd4, abb4, nz[u] * abc.strength, accb4, accd4, cs);
                    if (row_sums) {             // the 16 lanes of a DPP row hold the 64 channels of this pixel
                        cs = eg3d_row_group_sum(cs, 16);
                        if (c4 == 0) {
                            if (ab.dnoise != nullptr) eg3d_acc(ab.dnoise + (int64_t)n * ab.dnoise_nstride + pixl[u], cs * abc.strength);
                            accs += cs * nz[u];
                        }
                    }
                }
            }
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            *reinterpret_cast<float4*>(p.out + offs[u]) = v;
        }
    }
    if (do_ds || act_on) {
        if (do_ds) {
            [[maybe_unused]] float* gds = p.ds + (int64_t)n * p.Nc + col;
            EG3D_LDS_ACC(&ds_lds[c4 * 4 + 0], gds + 0, dsum4.x); EG3D_LDS_ACC(&ds_lds[c4 * 4 + 1], gds + 1, dsum4.y);
            EG3D_LDS_ACC(&ds_lds[c4 * 4 + 2], gds + 2, dsum4.z); EG3D_LDS_ACC(&ds_lds[c4 * 4 + 3], gds + 3, dsum4.w);
        }
        if (act_on) {
            if (ab.dbias != nullptr) {
                [[maybe_unused]] float* gdb = ab.dbias + col;
                EG3D_LDS_ACC(&db_lds[c4 * 4 + 0], gdb + 0, accb4.x); EG3D_LDS_ACC(&db_lds[c4 * 4 + 1], gdb + 1, accb4.y);
                EG3D_LDS_ACC(&db_lds[c4 * 4 + 2], gdb + 2, accb4.z); EG3D_LDS_ACC(&db_lds[c4 * 4 + 3], gdb + 3, accb4.w);
            }
            if (ab.dd != nullptr) {
                [[maybe_unused]] float* gdq = ab.dd + (int64_t)n * p.Nc + col;
                EG3D_LDS_ACC(&dq_lds[c4 * 4 + 0], gdq + 0, EG3D_DET_DIV(accd4.x, abd4.x)); EG3D_LDS_ACC(&dq_lds[c4 * 4 + 1], gdq + 1, EG3D_DET_DIV(accd4.y, abd4.y));
                EG3D_LDS_ACC(&dq_lds[c4 * 4 + 2], gdq + 2, EG3D_DET_DIV(accd4.z, abd4.z)); EG3D_LDS_ACC(&dq_lds[c4 * 4 + 3], gdq + 3, EG3D_DET_DIV(accd4.w, abd4.w));
            }
            if (ab.dstrength != nullptr && accs != 0.f) EG3D_LDS_ACC(sc_lds, ab.dstrength, accs);
        }
        __syncthreads();
#if !EG3D_DET
        if (tid < BN3) {
            if (do_ds) eg3d_acc(p.ds + (int64_t)n * p.Nc + n0 + tid, ds_lds[tid]);
            if (act_on && ab.dbias != nullptr) eg3d_acc(ab.dbias + n0 + tid, db_lds[tid]);
            if (act_on && ab.dd != nullptr)       // dL/dd = sum dy * z,  z = (pre - bias - noise) / d
                eg3d_acc(ab.dd + (int64_t)n * p.Nc + n0 + tid, dq_lds[tid] / (ab.d != nullptr ? ab.d[(int64_t)n * p.Nc + n0 + tid] : 1.f));
        }
        if (act_on && ab.dstrength != nullptr && tid == 0 && sc_lds[0] != 0.f) eg3d_acc(ab.dstrength, sc_lds[0]);
#endif
    }
    eg3d_commit_amax_block(amax, p.out_amax);
}

// ---- after the main loop: the NW partial tiles meet in LDS, the sum is staged as [cell][64] and the fused epilogue runs on it ---------------
template <int RPW, int NW>
__device__ __forceinline__ void v3_finish(const eg3d_conv_v2_params& p, f32x16 (&acc)[RPW][2], char* smem, const int wave, const int lane, const int Ha, const int Wa,
                                          const int out_py, const int out_px, const int n, const int y0, const int x0, const int n0) {
    constexpr int T = 2 * RPW;                            // 32 x 32 accumulator tiles of the workgroup tile
    constexpr int OWN = NW < T ? NW : T;                  // waves that own tiles after the reduction
    constexpr int TPO = T / OWN;                          // tiles per owner
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- reduction over the NW K slices: tile t = 2 i + j belongs to wave t / TPO; the others hand their partial tile over through LDS ------
    const float out_mul = 1.f / (*p.a_scale * *p.w_scale);
    {
        f32x4v* part = reinterpret_cast<f32x4v*>(smem);               // block (t, s') = 16 x 64 floats as [r / 4][lane][4]
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t = 2 * i + j, o = t / TPO;
                if (wave != o) {
                    const int sp = wave < o ? wave : wave - 1;
                    f32x4v* dst = part + (t * (NW - 1) + sp) * 256 + lane;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q * 64] = f32x4v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
            }
        __syncthreads();
        if (wave < OWN) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int t = 2 * i + j;
                    if (t / TPO != wave) continue;
                    f32x16 sum;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum[r] = 0.f;
                    bool first = true;
#pragma unroll
                    for (int s = 0; s < NW; ++s) {                 // ascending slice order, whoever owns the tile
                        f32x16 v;
                        if (s == wave) {
                            v = acc[i][j];
                        } else {
                            const int sp = s < wave ? s : s - 1;
                            const f32x4v* src = part + (t * (NW - 1) + sp) * 256 + lane;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4v w4 = src[q * 64];
                                v[4 * q] = w4[0]; v[4 * q + 1] = w4[1]; v[4 * q + 2] = w4[2]; v[4 * q + 3] = w4[3];
                            }
                        }
                        if (first) { sum = v; first = false; }
                        else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) sum[r] += v[r];
                        }
                    }
                    acc[i][j] = sum;
                }
        }
        __syncthreads();
        float* stage = reinterpret_cast<float*>(smem);
        if (wave < OWN) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if ((2 * i + j) / TPO != wave) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * STG_N + j * 32 + (lane & 31)] = acc[i][j][r] * out_mul;
                }
        }
        __syncthreads();
        v3_epilogue<RPW, NW>(p, stage, Ha, Wa, out_py, out_px, n, y0, x0, n0);
    }
}


// FULL: three products per fp32 product; !FULL: high pieces only (EG3D_PREC_F16X1).  RPW: rows of 32 cells per tile (4 | 2).  NW: waves = K slices.
template <bool FULL, int RPW, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 4) conv_v3_kernel(const eg3d_conv_v2_params p, const int cls_base) {
    using G = v3g<RPW, FULL>;
    constexpr int NTAPS = 9;
    constexpr int NB = FULL ? 4 : 2;                      // B loads per step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const eg3d_conv_class& cl = p.cls[cls_base + blockIdx.z];
    const int Ha = cl.Ha, Wa = cl.Wa;
    const int tiles_x = (Wa + PW - 1) / PW, tiles_y = (Ha + RPW - 1) / RPW, ntile_n = p.Nc / BN3;
    const int ntile = p.N * tiles_y * tiles_x * ntile_n;
    int bid = blockIdx.x;
    if (bid >= ntile) return;
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int n = bid / tiles_y;
    const int y0 = ty * RPW, x0 = tx * PW, n0 = n_t * BN3;
    const int nchunk = p.Ck / 16;
    const int planeA = p.Hi * p.Wi * 16;
    int dymin = cl.dy[0], dxmin = cl.dx[0];
#pragma unroll
    for (int t = 1; t < NTAPS; ++t) { dymin = min(dymin, cl.dy[t]); dxmin = min(dxmin, cl.dx[t]); }
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned wlds = (unsigned)(wave * G::WAVE_LDS);

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * planeA), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;

    // per-tap constants live in the lanes of two VGPRs (v_readlane with a constant lane): as kernel-argument table entries they were scalar loads
    // inside the loop, and every s_waitcnt lgkmcnt(0) for one also drained the LDS reads in flight
    int tapoff_v = 0, tapw_v = 0;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        if (lane == t) { tapoff_v = (cl.dy[t] * HW3 + cl.dx[t]) * 16; tapw_v = cl.wtap[t] * nchunk * 4 * p.Nc * 16; }
    }
    // ---- A loader: DMA op k of a chunk covers slots 64 k .. 64 k + 63 of the buffer's plane run; a_src[k] = this lane's source offset inside
    //      the (image, chunk) block: pixel + (piece, k-octet) plane -- chunk-independent; the chunk's block offset goes into the scalar offset ----
    unsigned a_src[G::NOPS_USED];
#pragma unroll
    for (int k = 0; k < G::NOPS_USED; ++k) {
        const int gs = k * 64 + lane;
        const int plane = gs / G::SLOTS, slot = gs - plane * G::SLOTS;
        const int hy = slot / HW3, hx = slot - hy * HW3;
        const int y = y0 + dymin + hy, x = x0 + dxmin + hx;
        const bool ok = plane < (FULL ? 4 : 2) && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        a_src[k] = ok ? (unsigned)((y * p.Wi + x) * 16 + ((plane >> 1) * (p.Ck / 8) + (plane & 1)) * planeA) : OOB;   // OOB + block offset stays out of bounds (no wrap): zeros
    }
    const int a_img = n * 2 * (p.Ck / 8);
    auto issue_A = [&](int chunk, int buf, int k) {
        glds16s(ars, lds0 + wlds + buf * G::ABUF3 + k * 1024, a_src[k], (a_img + chunk * 2) * planeA);
    };
    // ---- B loader: registers, ring of three steps --------------------------------------------------------------------------------------
    const unsigned b_lane = (unsigned)(((lane >> 5) * p.Nc + n0 + (lane & 31)) * 16);
    const int b_chunk = 4 * p.Nc * 16, b_piece = 2 * p.Nc * 16;
    u32x4 breg[3][NB];
    auto issue_B = [&](int chunk, int tap, int slot) {
        const int sbase = __builtin_amdgcn_readlane(tapw_v, tap) + chunk * b_chunk;
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int piece = e >> 1, j = e & 1;
            breg[slot][e] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_lane + j * 512, sbase + piece * b_piece, 0);
        }
    };

    f32x16 acc[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned a_lane = wlds + (unsigned)(((0 - dymin) * HW3 + (lane & 31) - dxmin) * 16 + (lane >> 5) * G::APL);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
    // A fragments one step ahead: af[parity of the step][row][piece]
    f16x8 af[2][RPW][FULL ? 2 : 1];
    auto load_A = [&](int par, int buf, int tap) {
        const unsigned abase = a_lane + (unsigned)(buf * G::ABUF3) + (unsigned)__builtin_amdgcn_readlane(tapoff_v, tap);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            af[par][i][0] = *reinterpret_cast<const f16x8*>(smem + abase + i * HW3 * 16);
            if constexpr (FULL) af[par][i][1] = *reinterpret_cast<const f16x8*>(smem + abase + i * HW3 * 16 + 2 * G::APL);
        }
    };
    constexpr int DPT = (G::NOPS_USED + 5) / 6;           // DMA ops per tap during taps 0 .. 5 (3 | 2): the halo has two steps to land

    // chunks of this wave: wave, wave + NW, ...
    const int nmine = nchunk > wave ? (nchunk - wave + NW - 1) / NW : 0;
    if (nmine > 0) {
        // ---- prologue: A(first chunk), B(step 0), B(step 1), fragments of step 0 ----------------------------------------------------------
#pragma unroll
        for (int k = 0; k < G::NOPS_USED; ++k) issue_A(wave, 0, k);
        issue_B(wave, 0, 0);
        issue_B(wave, 1, 1);
        wait_vm3<2 * NB>();
        asm volatile("" ::: "memory");
        load_A(0, 0, 0);
        auto run_chunk = [&](const int chunk, const int buf, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) {
                if constexpr (!LAST) {
                    if (tap < 6) {
#pragma unroll
                        for (int e = 0; e < DPT; ++e)
                            if (tap * DPT + e < G::NOPS_USED) issue_A(chunk + NW, buf ^ 1, tap * DPT + e);
                    }
                    if (tap == NTAPS - 1) {                      // the next chunk's halo: every DMA op is older than the last NB loads (B of its tap 0)
                        wait_vm3<NB>();
                        asm volatile("" ::: "memory");
                    }
                }
                if (tap + 2 < NTAPS) issue_B(chunk, tap + 2, (tap + 2) % 3);
                else if constexpr (!LAST) issue_B(chunk + NW, tap + 2 - NTAPS, (tap + 2) % 3);
                if (tap + 1 < NTAPS) load_A((tap + 1) & 1, buf, tap + 1);
                else if constexpr (!LAST) load_A(1, buf ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0);               // the loads above are issued BEFORE this step's MFMAs: a whole step to land
                f16x8 bh[2], bl[2], bg[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = as_f16x8(breg[tap % 3][j]);
                    if constexpr (FULL) {
                        bl[j] = as_f16x8(breg[tap % 3][2 + j]);
                        f16x2* s2 = reinterpret_cast<f16x2*>(&bh[j]);
                        f16x2* d2 = reinterpret_cast<f16x2*>(&bg[j]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) d2[q] = s2[q] * k2m11;
                    }
                }
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const f16x8 ah = af[tap & 1][i][0];
                    if constexpr (FULL) {
                        const f16x8 al = af[tap & 1][i][1];
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
            if constexpr (!LAST) {                               // step parity restarts with every chunk (nine taps)
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int q = 0; q < (FULL ? 2 : 1); ++q) af[0][i][q] = af[1][i][q];
            }
            asm volatile("" ::: "memory");                       // the next chunk's DMA issue stays behind this chunk's LDS reads
        };
        int chunk = wave, buf = 0;
        for (int k = 0; k + 1 < nmine; ++k, chunk += NW, buf ^= 1) run_chunk(chunk, buf, std::false_type{});
        run_chunk(chunk, buf, std::true_type{});
    }
    v3_finish<RPW, NW>(p, acc, smem, wave, lane, Ha, Wa, cl.out_py, cl.out_px, n, y0, x0, n0);
}

// ---- data gradient of the up-sampling layers' transposed convolution (a stride-2 3x3 correlation; conv_v2_s2adj.hip's contract) -----------------
//     dx[n, a, b, ci] = sum_{ky,kx} sum_co  g[n, 2a + ky, 2b + kx, co] * w[co, ci, ky, kx]
// on the PARITY-split image of g (eg3d_fir44_adjoint_split): tap (ky, kx) reads parity (ky & 1, kx & 1) at offset (ky >> 1, kx >> 1).  The
// contraction is the list of ITEMS (parity q, 16-channel chunk c) with 4 / 2 / 2 / 1 taps; the four waves of a workgroup take contiguous runs of
// that list of equal cost (9 nchunk / 4 tap-steps each) and every wave accumulates the WHOLE 128-cell x 64-channel tile for its items -- the
// decomposition of conv_v3_kernel with items in place of chunks.  An item's halo is 5 x 33 cells of ONE parity image (10.3 KB for the four
// (piece, k-octet) planes): private, double-buffered, by LDS-DMA; its weight tiles go straight into registers (two sets).  Loads are pipelined
// per item: wait for everything issued during the previous item, issue the next item's, multiply.  No barrier in the loop; v3_finish reduces.
struct v3a {
    static constexpr int RPW = 4, NW = 4;
    static constexpr int HWA = PW + 1;                          // halo pitch: 33 cells
    static constexpr int SLOTS = (RPW + 1) * HWA;              // 165
    static constexpr int APL = SLOTS * 16;
    static constexpr int NOPS = (4 * SLOTS + 63) / 64;         // 11
    static constexpr int ABUF3 = NOPS * 1024;
    static constexpr int WAVE_LDS = 2 * ABUF3;                 // 22528
};
constexpr int v3a_lds_bytes() {
    constexpr int main_b = v3a::NW * v3a::WAVE_LDS, red_b = 2 * v3a::RPW * (v3a::NW - 1) * 4096;
    return main_b > red_b ? main_b : red_b;
}
// taps of a kind (parity 2 py + px) as 3 ky + kx, and their halo offsets (ky >> 1, kx >> 1)
constexpr int v3a_nt(int q) { return q == 0 ? 4 : (q == 3 ? 1 : 2); }
constexpr int v3a_tap(int q, int j) { return q == 0 ? (j == 0 ? 0 : (j == 1 ? 2 : (j == 2 ? 6 : 8))) : (q == 1 ? (j == 0 ? 1 : 7) : (q == 2 ? (j == 0 ? 3 : 5) : 4)); }

template <bool FULL>
__global__ void __launch_bounds__(256, 1) conv_v3_s2adj_kernel(const eg3d_conv_v2_params p) {
    using G = v3a;
    constexpr int RPW = G::RPW, NW = G::NW, NB = FULL ? 4 : 2;
    constexpr int NOPS_USED = FULL ? G::NOPS : (2 * G::SLOTS + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const eg3d_conv_class& cl = p.cls[0];
    const int Ha = cl.Ha, Wa = cl.Wa;
    const int tiles_x = (Wa + PW - 1) / PW, tiles_y = (Ha + RPW - 1) / RPW, ntile_n = p.Nc / BN3;
    int bid = blockIdx.x;
    if (bid >= p.N * tiles_y * tiles_x * ntile_n) return;
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int n = bid / tiles_y;
    const int y0 = ty * RPW, x0 = tx * PW, n0 = n_t * BN3;
    const int nchunk = p.Ck / 16;
    const int Hp = p.Hi, Wp = p.Wi;
    const int planeP = Hp * Wp * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned wlds = (unsigned)(wave * G::WAVE_LDS);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * 4 * planeP), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;
    unsigned a_src[NOPS_USED];
#pragma unroll
    for (int k = 0; k < NOPS_USED; ++k) {
        const int gs = k * 64 + lane;
        const int plane = gs / G::SLOTS, slot = gs - plane * G::SLOTS;
        const int hy = slot / G::HWA, hx = slot - hy * G::HWA;
        const int y = y0 + hy, x = x0 + hx;
        const bool ok = plane < (FULL ? 4 : 2) && y < Hp && x < Wp;            // cells a parity image does not have hold zeros (eg3d_fir44_adjoint_split)
        a_src[k] = ok ? (unsigned)((y * Wp + x) * 16 + ((plane >> 1) * (p.Ck / 8) + (plane & 1)) * 4 * planeP) : OOB;
    }
    const int a_img = n * 2 * (p.Ck / 8);
    auto issue_A = [&](int q, int chunk, int buf) {
        const int soff = ((a_img + chunk * 2) * 4 + q) * planeP;
#pragma unroll
        for (int k = 0; k < NOPS_USED; ++k) glds16s(ars, lds0 + wlds + buf * G::ABUF3 + k * 1024, a_src[k], soff);
    };
    const unsigned b_lane = (unsigned)(((lane >> 5) * p.Nc + n0 + (lane & 31)) * 16);
    const int b_chunk = 4 * p.Nc * 16, b_piece = 2 * p.Nc * 16, b_tap = nchunk * b_chunk;
    const int wt0 = cl.wtap[0] * b_tap, wt1 = cl.wtap[1] * b_tap, wt2 = cl.wtap[2] * b_tap, wt3 = cl.wtap[3] * b_tap, wt4 = cl.wtap[4] * b_tap,
              wt5 = cl.wtap[5] * b_tap, wt6 = cl.wtap[6] * b_tap, wt7 = cl.wtap[7] * b_tap, wt8 = cl.wtap[8] * b_tap;
    u32x4 breg[2][4][NB];
    f32x16 acc[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned a_lane = wlds + (unsigned)((lane & 31) * 16 + (lane >> 5) * G::APL);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
    // this wave's run of the item list: items (q, c) q-major, start cost B_q + c cost_q, owner = floor(4 start / (9 nchunk)).  Smallest c with
    // 4 (base + c cost) >= w T9 (ceil division, clamped to [0, nchunk]):
    const int T9 = 9 * nchunk;
    auto first_c = [&](int w, int base, int cost) { const int num = w * T9 - 4 * base, den = 4 * cost; const int cc = num <= 0 ? 0 : (num + den - 1) / den; return cc > nchunk ? nchunk : cc; };
    // One loop per kind, each specialised at compile time (a run-time branch around an MFMA block makes the compiler shuttle the accumulators
    // between register files at every join); loads are pipelined inside a kind, a kind starts cold (three exposed latencies per wave at most).
    auto run_kind = [&](auto q_tag, const int lo, const int hi) {
        constexpr int Q = decltype(q_tag)::value;
        constexpr int NT = v3a_nt(Q);
        auto issue_B = [&](int chunk, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            auto one = [&](auto j_tag) {
                constexpr int J = decltype(j_tag)::value;
                if constexpr (J < NT) {
                    constexpr int TAP = v3a_tap(Q, J);
                    const int wbase = (TAP == 0 ? wt0 : TAP == 1 ? wt1 : TAP == 2 ? wt2 : TAP == 3 ? wt3 : TAP == 4 ? wt4 : TAP == 5 ? wt5 : TAP == 6 ? wt6 : TAP == 7 ? wt7 : wt8) + chunk * b_chunk;
#pragma unroll
                    for (int e = 0; e < NB; ++e)
                        breg[SET][J][e] = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_lane + (e & 1) * 512, wbase + (e >> 1) * b_piece, 0);
                }
            };
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
        };
        auto item = [&](const int chunk, const bool has_next, auto set_tag) {
            constexpr int SET = decltype(set_tag)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this item's halo and weight tiles
            if (has_next) {
                issue_A(Q, chunk + 1, SET ^ 1);
                issue_B(chunk + 1, std::integral_constant<int, SET ^ 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            auto one = [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                if constexpr (j < NT) {
                    constexpr int t = v3a_tap(Q, j);
                    constexpr int sy = (t / 3) >> 1, sx = (t % 3) >> 1;
                    const unsigned abase = a_lane + (unsigned)(SET * G::ABUF3 + (sy * G::HWA + sx) * 16);
                    f16x8 bh[2], bl[2], bg[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        bh[jj] = as_f16x8(breg[SET][j][jj]);
                        if constexpr (FULL) {
                            bl[jj] = as_f16x8(breg[SET][j][2 + jj]);
                            f16x2* s2 = reinterpret_cast<f16x2*>(&bh[jj]);
                            f16x2* d2 = reinterpret_cast<f16x2*>(&bg[jj]);
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) d2[qq] = s2[qq] * k2m11;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
                        const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + abase + i * G::HWA * 16);
                        if constexpr (FULL) {
                            const f16x8 al = *reinterpret_cast<const f16x8*>(smem + abase + i * G::HWA * 16 + 2 * G::APL);
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bg[jj], acc[i][jj], 0, 0, 0);
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[jj], acc[i][jj], 0, 0, 0);
                                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[jj], acc[i][jj], 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[jj], acc[i][jj], 0, 0, 0);
                        }
                    }
                }
            };
            one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
            asm volatile("" ::: "memory");
        };
        const int cnt = hi - lo;
        if (cnt > 0) {
            issue_A(Q, lo, 0);
            issue_B(lo, std::integral_constant<int, 0>{});
            int k = 0;
            for (; k + 1 < cnt; k += 2) {
                item(lo + k, true, std::integral_constant<int, 0>{});
                item(lo + k + 1, k + 2 < cnt, std::integral_constant<int, 1>{});
            }
            if (k < cnt) item(lo + k, false, std::integral_constant<int, 0>{});
        }
    };
    run_kind(std::integral_constant<int, 0>{}, first_c(wave, 0, 4), first_c(wave + 1, 0, 4));
    run_kind(std::integral_constant<int, 1>{}, first_c(wave, 4 * nchunk, 2), first_c(wave + 1, 4 * nchunk, 2));
    run_kind(std::integral_constant<int, 2>{}, first_c(wave, 6 * nchunk, 2), first_c(wave + 1, 6 * nchunk, 2));
    run_kind(std::integral_constant<int, 3>{}, first_c(wave, 8 * nchunk, 1), first_c(wave + 1, 8 * nchunk, 1));
    v3_finish<RPW, NW>(p, acc, smem, wave, lane, Ha, Wa, cl.out_py, cl.out_px, n, y0, x0, n0);
}

std::atomic<uint64_t> g_attr3[10];

template <bool FULL, int RPW, int NW>
int launch_v3(const eg3d_conv_v2_params& p, int cls_base, int ncls, int max_tiles, hipStream_t st, int slot) {
    auto kern = conv_v3_kernel<FULL, RPW, NW>;
    constexpr int lds = v3_lds_bytes<RPW, NW>();
    static_assert(lds <= 160 * 1024 - 2048, "LDS budget (static epilogue arrays included)");
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, g_attr3[slot])) return e;
    hipLaunchKernelGGL(kern, dim3(max_tiles, 1, ncls), dim3(NW * 64), lds, st, p, cls_base);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

}  // namespace

extern "C" int eg3d_conv2d_v3_supported(const eg3d_conv_v2_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_v2_params& p = *pp;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck < 16 || (p.Ck & 15) || p.Nc < BN3 || (p.Nc % BN3) || (p.ldo & 3)) return 0;
    if (p.in_stride != 1 || p.out_stride < 1 || p.ncls < 1 || p.ncls > 4) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_FWD && p.epi != EG3D_EPI_BWD && p.epi != EG3D_EPI_BWD_ACT) return 0;
    if (p.patch_rows != 0 && p.patch_rows != 4 && p.patch_rows != 2) return 0;
    if (p.ksplit != 0 && p.ksplit != 4 && p.ksplit != 8) return 0;
    if (p.ksplit == 8 && p.patch_rows != 2) return 0;     // eight private halos of a 4-row patch do not fit the LDS
    if (p.epi == EG3D_EPI_FWD && !eg3d_act_is_pwl(p.act)) return 0;
    if (p.epi == EG3D_EPI_BWD_ACT) {
        const eg3d_act_bwd& ab = p.act_bwd;
        if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return 0;
        if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return 0;
    }
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.ntaps != 9) return 0;
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

extern "C" int eg3d_conv2d_v3(const eg3d_conv_v2_params* pp, void* stream) {
    if (!pp || !pp->a || !pp->w || !pp->out || !pp->a_scale || !pp->w_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_v3_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_v2_params& p = *pp;
    if (p.epi == EG3D_EPI_BWD_ACT && !p.xin) return EG3D_ERR_INVALID;
    const void* ptrs[] = {p.out, p.addend, p.xin, p.out_scale, p.bias, p.act_bwd.d, p.act_bwd.bias};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.epi == EG3D_EPI_FWD && p.noise && !p.noise_strength) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND_V2(det, p); EG3D_DET_COMMIT(det);
    const int rpw = p.patch_rows == 2 ? 2 : 4, nw = p.ksplit == 8 ? 8 : 4;
    const bool full = p.products != 1;
    int max_tiles = 0;
    for (int c = 0; c < p.ncls; ++c)
        max_tiles = std::max(max_tiles, p.N * eg3d_cdiv(p.cls[c].Ha, rpw) * eg3d_cdiv(p.cls[c].Wa, PW) * (p.Nc / BN3));
    int rc;
    if (rpw == 4) rc = full ? launch_v3<true, 4, 4>(p, 0, p.ncls, max_tiles, st, 0) : launch_v3<false, 4, 4>(p, 0, p.ncls, max_tiles, st, 1);
    else if (nw == 4) rc = full ? launch_v3<true, 2, 4>(p, 0, p.ncls, max_tiles, st, 2) : launch_v3<false, 2, 4>(p, 0, p.ncls, max_tiles, st, 3);
    else rc = full ? launch_v3<true, 2, 8>(p, 0, p.ncls, max_tiles, st, 4) : launch_v3<false, 2, 8>(p, 0, p.ncls, max_tiles, st, 5);
    if (rc != EG3D_OK) return rc;
    EG3D_DET_END(det);
    return EG3D_OK;
}

extern "C" int eg3d_conv2d_v3_s2adj_supported(const eg3d_conv_v2_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_v2_params& p = *pp;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck < 16 || (p.Ck & 15) || p.Nc < BN3 || (p.Nc % BN3) || (p.ldo & 3)) return 0;
    if (p.out_stride != 1 || p.ncls != 1 || p.wtaps < 9) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_BWD && p.epi != EG3D_EPI_BWD_ACT) return 0;
    if (p.epi == EG3D_EPI_BWD_ACT) {
        const eg3d_act_bwd& ab = p.act_bwd;
        if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return 0;
        if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return 0;
    }
    const eg3d_conv_class& k = p.cls[0];
    if (k.ntaps != 9 || k.out_py != 0 || k.out_px != 0) return 0;
    for (int t = 0; t < 9; ++t)
        if (k.dy[t] != t / 3 || k.dx[t] != t % 3 || k.wtap[t] < 0 || k.wtap[t] >= p.wtaps) return 0;
    if (p.Hi < k.Ha + 1 || p.Wi < k.Wa + 1) return 0;         // parity images cover the class grid + the offset-1 taps
    if ((int64_t)p.N * 2 * (p.Ck / 8) * 4 * p.Hi * p.Wi * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.N * p.Ho * p.Wo * p.ldo > INT32_MAX) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_v3_s2adj(const eg3d_conv_v2_params* pp, void* stream) {
    if (!pp || !pp->a || !pp->w || !pp->out || !pp->a_scale || !pp->w_scale) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_v3_s2adj_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_v2_params& p = *pp;
    if (p.epi == EG3D_EPI_BWD_ACT && !p.xin) return EG3D_ERR_INVALID;
    const void* ptrs[] = {p.out, p.addend, p.xin, p.out_scale, p.act_bwd.d, p.act_bwd.bias};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND_V2(det, p); EG3D_DET_COMMIT(det);
    const int tiles = p.N * eg3d_cdiv(p.cls[0].Ha, 4) * eg3d_cdiv(p.cls[0].Wa, PW) * (p.Nc / BN3);
    constexpr int lds = v3a_lds_bytes();
    if (p.products != 1) {
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_v3_s2adj_kernel<true>), lds, g_attr3[8])) return e;
        hipLaunchKernelGGL(conv_v3_s2adj_kernel<true>, dim3(tiles), dim3(256), lds, st, p);
    } else {
        if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_v3_s2adj_kernel<false>), lds, g_attr3[9])) return e;
        hipLaunchKernelGGL(conv_v3_s2adj_kernel<false>, dim3(tiles), dim3(256), lds, st, p);
    }
    EG3D_LAUNCH_CHECK();
    EG3D_DET_END(det);
    return EG3D_OK;
}
