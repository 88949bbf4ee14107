// Shared pieces of the pre-split convolution kernels (conv_v2.hip: stride-1 tap classes; conv_v2_s2adj.hip: the stride-2 adjoint of the
// up-sampling layers): LDS geometry of the 8 x 32 patch x 128 channel tile, the LDS-DMA helper, counted vmcnt waits and the fused epilogue.
#pragma once
#include "common.h"
#include "det.h"
// accumulation targets of a launch on eg3d_conv_v2_params (split-K output, style gradient, the producer's activation backward)
#define EG3D_DET_BIND_V2(name, p) do { \
        if ((p).epi == EG3D_EPI_ATOMIC) EG3D_DET_BIND(name, (p).out, (int64_t)(p).N * (p).Ho * (p).Wo * (p).ldo); \
        EG3D_DET_BIND(name, (p).ds, (int64_t)(p).N * (p).Nc); \
        if ((p).epi == EG3D_EPI_BWD_ACT) EG3D_DET_BIND_ACT(name, (p).act_bwd, (p).N, (p).Nc, (int64_t)(p).Ho * (p).Wo); \
    } while (0)

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


// ---- epilogue of a 256-cell x 128-channel tile held as acc[4][2] (wave (wm, wn): patch rows 4 wm .. 4 wm + 3, channels 64 wn .. + 63) ---
// Output cell (ay, ax) of the class grid Ha x Wa goes to pixel (ay * out_stride + out_py, ax * out_stride + out_px).  Must be entered by
// the whole block after the last LDS read of the main loop (it re-uses the dynamic LDS).
// GENW (tools/proto/conv_lr.hip, not built): the workgroup's cells are a (64 RPW >> logw) x (1 << logw) patch (narrow images: 16 / 8 / 4 columns) instead of rows of 32:
// cell c of MFMA tile mt is patch position lin = 32 mt + c -> row lin >> logw, column lin & ((1 << logw) - 1).  out_mul: 1 / (a_scale * w_scale).
template <bool ATOMIC, int RPW = 4, bool GENW = false, bool RGB = false>   // ATOMIC at compile time: the split-K form must not cost the fused epilogues a register (together they spilled 30 VGPRs);
                                      // RPW = patch rows per wave (4: 8 x 32 patch, 2: 4 x 32)
__device__ __forceinline__ void v2_epilogue(const eg3d_conv_v2_params& p, f32x16 (&acc)[RPW][2], const int Ha, const int Wa, const int out_py, const int out_px,
                                            const int n, const int y0, const int x0, const int n0, char* smem, const float out_mul, const int logw = 5,
                                            const int nwaves = 0 /* waves that enter (0 = the whole block) */) {
    const int wmask = (1 << logw) - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    struct { int out_py, out_px; } cl = {out_py, out_px};
    // ---- epilogue: the tile goes through LDS once (64 rows at a time) so that every global access is 16 bytes per lane ----------------
    const int epi = p.epi;
    if constexpr (ATOMIC) {
        // split-K partial tile: atomics straight from the accumulator layout, one lane per channel -- a wave-instruction covers two runs of
        // 32 consecutive floats.  (Through the staged float4 path each lane would own 4 consecutive channels: four instructions that each
        // touch sixteen 64-byte lines -- measured 2.3x SLOWER than the loader-split kernel on the 128^2 x 256 layer.)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cc = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ay = GENW ? y0 + (((wm * RPW + i) * 32 + cc) >> logw) : y0 + wm * RPW + i;
                const int ax = GENW ? x0 + (cc & wmask) : x0 + cc;
                if (ay >= Ha || ax >= Wa) continue;
                float* o = p.out + ((int64_t)(n * p.Ho + ay * p.out_stride + cl.out_py) * p.Wo + ax * p.out_stride + cl.out_px) * p.ldo + n0 + wn * 64 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 2; ++j) eg3d_acc(o + j * 32, acc[i][j][r] * out_mul);
            }
        }
    } else {
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
    // the 1x1 head (eg3d_conv_v2_params::rgb_out): the four modulated weight rows of this channel tile live in LDS behind the epilogue's staging
    // area (as 16 registers per thread across the whole epilogue they pushed the instantiation to 256 VGPRs + scratch: +25 us on a 200 us launch)
    const bool rgb_on = RGB && epi == EG3D_EPI_FWD && p.rgb_out != nullptr;          // (RGB: its own instantiation -- the others must not pay for it)
    float* rwl = reinterpret_cast<float*>(smem + ((LDS_EPI + 15) & ~15));             // [4][BN]
    float4 rgb_b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rgb_on) {
        if (tid < BN) {
            const float sv = p.rgb_s[(int64_t)n * p.Nc + n0 + tid];
#pragma unroll
            for (int o = 0; o < 4; ++o) rwl[o * BN + tid] = p.rgb_w[(int64_t)o * p.rgb_ldw + n0 + tid] * sv;
        }
        if (p.rgb_bias != nullptr) rgb_b = *reinterpret_cast<const float4*>(p.rgb_bias);
    }
    float4 abd4 = make_float4(1.f, 1.f, 1.f, 1.f), abb4 = make_float4(0.f, 0.f, 0.f, 0.f), accb4 = abb4, accd4 = abb4;
    float accs = 0.f;
    if (act_on && ab.d != nullptr) abd4 = *reinterpret_cast<const float4*>(ab.d + (int64_t)n * p.Nc + col);
    if (act_on && ab.bias != nullptr) abb4 = *reinterpret_cast<const float4*>(ab.bias + col);
    float4 rw[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) rw[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rgb_on) {           // three rows in registers (straight from global memory: 12 loads per thread, once), a fourth one only in LDS
        const float4 s4 = *reinterpret_cast<const float4*>(p.rgb_s + (int64_t)n * p.Nc + col);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float4 w4 = *reinterpret_cast<const float4*>(p.rgb_w + (int64_t)o * p.rgb_ldw + col);
            rw[o] = make_float4(w4.x * s4.x, w4.y * s4.y, w4.z * s4.z, w4.w * s4.w);
        }
    }
    float amax = 0.f;
    // the patch's noise values (forward: this layer's; EPI_BWD_ACT: the producing layer's), one per cell, read from LDS by the 32 lanes that
    // hold a cell's channels
    __shared__ float nzl[2 * 4 * 32];
    {
        const float* nsrc = (epi == EG3D_EPI_FWD && p.noise != nullptr) ? p.noise + (int64_t)n * p.noise_nstride
                          : ((act_on && ab.noise != nullptr) ? ab.noise + (int64_t)n * ab.noise_nstride : nullptr);
        if (tid < 2 * RPW * 32) {
            const int ay = GENW ? y0 + (tid >> logw) : y0 + (tid >> 5), ax = GENW ? x0 + (tid & wmask) : x0 + (tid & 31);
            float v = 0.f;
            if (nsrc != nullptr && ay < Ha && ax < Wa) v = nsrc[(ay * p.out_stride + cl.out_py) * p.Wo + ax * p.out_stride + cl.out_px];
            nzl[tid] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDS_N + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r] * out_mul;
        __syncthreads();
        // 64 rows x 32 float4 units = 2048 units, 8 per thread, in two groups of 4 (loads first, then arithmetic + stores); the instantiation
        // with the 1x1 head takes them in four groups of 2: the head's weight rows live in registers and the unit arrays make room for them
        constexpr int UGN = RGB ? 2 : 4;
#pragma unroll
        for (int ug = 0; ug < 8; ug += UGN) {
            int offs[UGN], pixl[UGN];
            float4 va[UGN], sa[UGN], sb[UGN];
            float nz[UGN];
#pragma unroll
            for (int u = 0; u < UGN; ++u) {
                const int row = (tid + (ug + u) * 256) >> 5;             // 0..63: wave-row row >> 5, patch column row & 31
                const int ay = GENW ? y0 + ((((row >> 5) * RPW + i) * 32 + (row & 31)) >> logw) : y0 + (row >> 5) * RPW + i;
                const int ax = GENW ? x0 + (row & wmask) : x0 + (row & 31);
                const bool ok = ay < Ha && ax < Wa;
                const int pix = (n * p.Ho + ay * p.out_stride + cl.out_py) * p.Wo + ax * p.out_stride + cl.out_px;
                offs[u] = ok ? pix * p.ldo + col : -1;
                pixl[u] = pix - n * HWo;
                va[u] = *reinterpret_cast<const float4*>(stage + row * LDS_N + c4 * 4);
                sa[u] = make_float4(0.f, 0.f, 0.f, 0.f); sb[u] = sa[u]; nz[u] = 0.f;
                if (ok && (epi == EG3D_EPI_FWD || bwd_like) && p.addend != nullptr) sa[u] = *reinterpret_cast<const float4*>(p.addend + offs[u]);
                nz[u] = nzl[((row >> 5) * RPW + i) * 32 + (row & 31)];
                if (ok && (do_ds || act_on)) sb[u] = *reinterpret_cast<const float4*>(p.xin + offs[u]);
            }
#pragma unroll
            for (int u = 0; u < UGN; ++u) {
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
                    if (rgb_on) {                   // the 32 lanes of a half-wave hold the 128 channels of this pixel
                        // four sums over the 32 lanes in 6 exchanges instead of 20: the first two steps halve the number of values a lane
                        // carries (lane bit 0 chooses outputs {0,1} | {2,3}, bit 1 the one of the pair), the rest are butterflies on ONE value
                        // that keep the low two lane bits (row_ror 4 / 8, then the other 16-lane row); lane l then holds output
                        // 2 (l & 1) + ((l >> 1) & 1) summed over all 32 lanes and lane 0 collects the four from its quad
                        const float q0 = fmaf(v.w, rw[0].w, fmaf(v.z, rw[0].z, fmaf(v.y, rw[0].y, v.x * rw[0].x)));
                        const float q1 = fmaf(v.w, rw[1].w, fmaf(v.z, rw[1].z, fmaf(v.y, rw[1].y, v.x * rw[1].x)));
                        const float q2 = fmaf(v.w, rw[2].w, fmaf(v.z, rw[2].z, fmaf(v.y, rw[2].y, v.x * rw[2].x)));
                        float q3 = 0.f;
                        if (p.rgb_nout != 3) {          // a real fourth output: its row stays in LDS
                            const float4 r = *reinterpret_cast<const float4*>(rwl + 3 * BN + c4 * 4);
                            q3 = fmaf(v.w, r.w, fmaf(v.z, r.z, fmaf(v.y, r.y, v.x * r.x)));
                        }
                        const bool b0 = (c4 & 1) != 0, b1 = (c4 & 2) != 0;
                        float ka = b0 ? q2 : q0, kb = b0 ? q3 : q1;                                 // what this lane keeps ...
                        ka += eg3d_dpp<0xB1>(b0 ? q0 : q2); kb += eg3d_dpp<0xB1>(b0 ? q1 : q3);     // ... plus what lane ^ 1 gives it (quad_perm [1,0,3,2])
                        float k = b1 ? kb : ka;
                        k += eg3d_dpp<0x4E>(b1 ? ka : kb);                                          // lane ^ 2 (quad_perm [2,3,0,1])
                        k += eg3d_dpp<0x124>(k);                                                    // row_ror 4
                        k += eg3d_dpp<0x128>(k);                                                    // row_ror 8
                        k += __shfl_xor(k, 16);
                        float r4[4];
                        r4[0] = k; r4[2] = eg3d_dpp<0x55>(k); r4[1] = eg3d_dpp<0xAA>(k); r4[3] = eg3d_dpp<0xFF>(k);      // quad broadcasts of lanes 1, 2, 3
                        if (c4 == 0) {
                            float4 y = make_float4(r4[0] + rgb_b.x, r4[1] + rgb_b.y, r4[2] + rgb_b.z, r4[3] + rgb_b.w);
                            if (p.rgb_clamp >= 0.f) {
                                y.x = fminf(fmaxf(y.x, -p.rgb_clamp), p.rgb_clamp); y.y = fminf(fmaxf(y.y, -p.rgb_clamp), p.rgb_clamp);
                                y.z = fminf(fmaxf(y.z, -p.rgb_clamp), p.rgb_clamp); y.w = fminf(fmaxf(y.w, -p.rgb_clamp), p.rgb_clamp);
                            }
                            *reinterpret_cast<float4*>(p.rgb_out + ((int64_t)n * HWo + pixl[u]) * 4) = y;
                        }
                    }
                } else if (bwd_like) {
                    if (do_ds) { dsum4.x += v.x * sb[u].x; dsum4.y += v.y * sb[u].y; dsum4.z += v.z * sb[u].z; dsum4.w += v.w * sb[u].w; }
                    v = make_float4(v.x * scl4.x + sa[u].x, v.y * scl4.y + sa[u].y, v.z * scl4.z + sa[u].z, v.w * scl4.w + sa[u].w);
                    if (act_on) {                   // v = dout of the layer that produced xin: its activation backward, here
                        float cs;
                        v = eg3d_act_bwd_unit(abc, v, sb[u], abd4, abb4, nz[u] * abc.strength, accb4, accd4, cs);
                        if (row_sums) {             // the 32 lanes of a half-wave hold the 128 channels of this pixel (rows are half-wave uniform)
                            cs = eg3d_row_group_sum(cs, 32);
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
        if (tid < BN) {
            if (do_ds) eg3d_acc(p.ds + (int64_t)n * p.Nc + n0 + tid, ds_lds[tid]);
            if (act_on && ab.dbias != nullptr) eg3d_acc(ab.dbias + n0 + tid, db_lds[tid]);
            if (act_on && ab.dd != nullptr)       // dL/dd = sum dy * z,  z = (pre - bias - noise) / d
                eg3d_acc(ab.dd + (int64_t)n * p.Nc + n0 + tid, dq_lds[tid] / (ab.d != nullptr ? ab.d[(int64_t)n * p.Nc + n0 + tid] : 1.f));
        }
        if (act_on && ab.dstrength != nullptr && tid == 0 && sc_lds[0] != 0.f) eg3d_acc(ab.dstrength, sc_lds[0]);
    }
    eg3d_commit_amax_block(amax, p.out_amax, nwaves);    // max|out|: the consumer's operand range (one atomic per block)
    }
}

}  // namespace
