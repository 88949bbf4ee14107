// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, 157 TF peak).
//
//   GEMM view:  M = output-grid cells (n,ay,ax)   N = output channels   K = taps x input channels
//   A[m][k]  gathered on the fly from the NHWC activation tensor (im2col never materialised), optionally scaled by the
//            per-(n,k) style (modulation folded into the operand load -> weights are shared by the whole batch);
//   B[n][k]  = w[o][tap][k]  (k contiguous, i.e. the channels_last image of a [O,I,kh,kw] weight).
//
//   Block = 256 threads = 4 waves (WM x WN); block tile BM x BN x 32; each wave owns (BM/WM) x (BN/WN) outputs as
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

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) conv_igemm_kernel(const eg3d_conv_params p) {
    constexpr int NT = WM * WN * 64;                        // 4 or 8 waves
    constexpr int RPP = NT / 8;                             // tile rows covered by one pass of the loader (8 float4 per row)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LD = BM / RPP, B_LD = BN / RPP;         // float4 per thread per tile
    static_assert(BM % RPP == 0 && BN % RPP == 0, "loader");
    static_assert(TM >= 1 && TN >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;          // [2][BN][LDK]
    int* rowpix = reinterpret_cast<int*>(Bs + 2 * BN * LDK);   // [BM] output pixel index (n*Ho+oy)*Wo+ox, -1 = none
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
    // ---- operand loaders: branch-free raw buffer loads ----------------------------------------------------------------
    // Every thread fetches A_LD + B_LD 16-byte pieces per K-step.  The per-row part of the address and a 9-bit "tap in bounds"
    // mask are computed once; per step the address is base + (wave-uniform tap/chunk offset), and an out-of-image tap (or
    // a row / channel past the end) is redirected to an out-of-range buffer offset, which the hardware returns as zeros.  No
    // divergent control flow in the K loop, so the loads interleave with the MFMAs of the previous step.
    const int lrow = tid >> 3, col4 = tid & 7;
    constexpr unsigned OOB = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)(((int64_t)(p.Nc - 1) * p.w_row + (int64_t)9 * p.Ck) * 4 > 0x7fffffe0 ? 0x7fffffe0 : ((int64_t)(p.Nc - 1) * p.w_row + (int64_t)9 * p.Ck) * 4), 0x00020000);
    unsigned a_base[A_LD], a_mask[A_LD];
    int a_n[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        int m = m0 + lrow + RPP * j;
        const bool ok = m < Mc;
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
        b_base[j] = row < p.Nc ? (unsigned)(((int64_t)row * p.w_row + col4 * 4) * 4) : OOB;
    }

    const int nchunks = (p.Ck + BK - 1) / BK;
    const int S = nchunks * ntaps;
    const int s_begin = (int)((int64_t)kslice * S / p.ksplit);
    const int s_end = (int)((int64_t)(kslice + 1) * S / p.ksplit);

    float4 ra[A_LD], rb[B_LD], sc[A_LD];
    int cur_chunk = -1;
    int nx_chunk = s_begin / ntaps, nx_tap = s_begin - nx_chunk * ntaps;      // (chunk, tap) of the next step to load

    auto load_tiles = [&]() {
        const int chunk = nx_chunk, tap = nx_tap;
        const bool kok = chunk * BK + col4 * 4 < p.Ck;
        if (p.in_scale != nullptr && chunk != cur_chunk) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j)
                sc[j] = kok ? *reinterpret_cast<const float4*>(p.in_scale + (int64_t)a_n[j] * p.Ck + chunk * BK + col4 * 4) : make_float4(0, 0, 0, 0);
            cur_chunk = chunk;
        }
        const unsigned aoff = (unsigned)(((cl.dy[tap] * p.Wi + cl.dx[tap]) * p.ldx + chunk * BK) * 4);       // wave-uniform
        const unsigned boff = (unsigned)((cl.wtap[tap] * p.Ck + chunk * BK) * 4);
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
    auto store_tiles = [&](int buf) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            float4 v = ra[j];
            if (p.in_scale != nullptr) v = f4mul(v, sc[j]);
            *reinterpret_cast<float4*>(a + (lrow + RPP * j) * LDK + col4 * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) *reinterpret_cast<float4*>(b + (lrow + RPP * j) * LDK + col4 * 4) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        load_tiles();
        store_tiles(0);
    }
    __syncthreads();

    const int arow = wm * (TM * 32) + (lane & 31);
    const int brow = wn * (TN * 32) + (lane & 31);
    const int khalf = (lane >> 5) * 4;

    for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        const bool more = step + 1 < s_end;
        if (more) load_tiles();
        const float* a = As + buf * BM * LDK;
        const float* b = Bs + buf * BN * LDK;
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
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------
    const int epi = p.epi;
    const int n_first = m0 / HWa;
    const int m_last = (m0 + BM < Mc ? m0 + BM : Mc) - 1;
    const bool single_n = (m_last / HWa) == n_first;
    float* ds_lds = As;                     // BN floats, reused after the final barrier
    const bool do_ds = (epi == EG3D_EPI_BWD) && p.ds != nullptr && p.xin != nullptr;
    if (do_ds && single_n) {
        if (tid < BN) ds_lds[tid] = 0.f;
        __syncthreads();
    }
    const float strength = (epi == EG3D_EPI_FWD && p.noise != nullptr) ? *p.noise_strength : 0.f;
    const int HWo = p.Ho * p.Wo;

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
        const bool cok = col < p.Nc;
        float dsum = 0.f;
        const float bias = (epi == EG3D_EPI_FWD && p.bias != nullptr && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int pix = rowpix[rl];
                if (pix < 0 || !cok) continue;
                const int64_t off = (int64_t)pix * p.ldo + col;
                float v = acc[i][j][r];
                if (epi == EG3D_EPI_STORE) {
                    p.out[off] = v;
                } else if (epi == EG3D_EPI_ATOMIC) {
                    unsafeAtomicAdd(p.out + off, v);
                } else if (epi == EG3D_EPI_FWD) {
                    const int n = rown[rl];
                    if (p.out_scale != nullptr) v *= p.out_scale[(int64_t)n * p.Nc + col];
                    if (p.noise != nullptr) v += p.noise[(int64_t)n * p.noise_nstride + (pix - n * HWo)] * strength;
                    v += bias;
                    v = eg3d_act_fwd<float>(v, p.act, p.alpha) * p.gain;
                    if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
                    if (p.addend != nullptr) v += p.addend[off];
                    p.out[off] = v;
                } else {   // EG3D_EPI_BWD
                    const int n = rown[rl];
                    if (do_ds) {
                        float t = v * p.xin[off];
                        if (single_n) dsum += t;
                        else unsafeAtomicAdd(p.ds + (int64_t)n * p.Nc + col, t);
                    }
                    if (p.out_scale != nullptr) v *= p.out_scale[(int64_t)n * p.Nc + col];
                    if (p.addend != nullptr) v += p.addend[off];
                    p.out[off] = v;
                }
            }
        }
        if (do_ds && single_n) {
            dsum += __shfl_xor(dsum, 32);
            if (lane < 32 && cok) atomicAdd(&ds_lds[wn * (TN * 32) + j * 32 + lane], dsum);
        }
    }
    if (do_ds && single_n) {
        __syncthreads();
        if (tid < BN && n0 + tid < p.Nc) unsafeAtomicAdd(p.ds + (int64_t)n_first * p.Nc + n0 + tid, ds_lds[tid]);
    }
}

template <int BM, int BN, int WM, int WN>
int launch_conv(const eg3d_conv_params& p, hipStream_t st) {
    static bool attr_done = false;
    const size_t smem = (size_t)(2 * (BM + BN) * LDK) * sizeof(float) + 2 * BM * sizeof(int);
    auto kern = conv_igemm_kernel<BM, BN, WM, WN>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
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

// tile configuration: 0 = 128x128 (dominant), 1 = 64x128, 2 = 32x128 (tiny spatial extent), 3 = 128x32 (few output channels)
int pick_config(const eg3d_conv_params& p) {
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) maxM = std::max<int64_t>(maxM, (int64_t)p.N * p.cls[c].Ha * p.cls[c].Wa);
    if (p.Nc <= 32) return 3;
    if (maxM <= 32) return 2;
    // enough 128x128 tiles to fill the chip (2 blocks/CU)?  otherwise shrink the M tile
    int64_t big_tiles = (int64_t)eg3d_cdiv(maxM, 128) * eg3d_cdiv(p.Nc, 128) * p.ncls * p.ksplit;
    if (big_tiles >= 384) return 0;
    if (maxM <= 64 * 8) return 2;
    return 1;
}

}  // namespace

extern "C" int eg3d_conv2d_igemm_f32(const eg3d_conv_params* pp, void* stream) {
    if (!pp) return EG3D_ERR_INVALID;
    const eg3d_conv_params& p = *pp;
    if (!p.x || !p.w || !p.out) return EG3D_ERR_INVALID;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck <= 0 || p.Nc <= 0 || p.Ho <= 0 || p.Wo <= 0) return EG3D_ERR_INVALID;
    if (p.ncls < 1 || p.ncls > 4 || p.ksplit < 1 || p.in_stride < 1 || p.out_stride < 1) return EG3D_ERR_INVALID;
    if (p.ksplit > 1 && p.epi != EG3D_EPI_ATOMIC) return EG3D_ERR_INVALID;
    if (p.epi < EG3D_EPI_STORE || p.epi > EG3D_EPI_BWD) return EG3D_ERR_INVALID;
    if (p.epi == EG3D_EPI_FWD && p.noise && !p.noise_strength) return EG3D_ERR_INVALID;
    if ((p.Ck & 3) || (p.ldx & 3) || (p.w_row & 3)) return EG3D_ERR_UNSUPPORTED;   // 16-byte operand loads
    if ((reinterpret_cast<uintptr_t>(p.x) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.in_scale && (reinterpret_cast<uintptr_t>(p.in_scale) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.ldx < p.Ck || p.ldo < p.Nc) return EG3D_ERR_INVALID;
    int64_t maxM = 0;
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.Ha <= 0 || k.Wa <= 0 || k.ntaps < 1 || k.ntaps > 9) return EG3D_ERR_INVALID;
        if ((k.Ha - 1) * p.out_stride + k.out_py >= p.Ho || (k.Wa - 1) * p.out_stride + k.out_px >= p.Wo) return EG3D_ERR_INVALID;
        maxM = std::max<int64_t>(maxM, (int64_t)p.N * k.Ha * k.Wa);
    }
    if ((int64_t)p.N * p.Hi * p.Wi * p.ldx * 4 > 0x7fffffe0ll || (int64_t)p.N * p.Ho * p.Wo * p.ldo > INT32_MAX) return EG3D_ERR_TOO_LARGE;
    if ((int64_t)p.Nc * p.w_row * 4 > 0x7fffffe0ll) return EG3D_ERR_TOO_LARGE;       // 31-bit buffer offsets
    hipStream_t st = (hipStream_t)stream;
    switch (pick_config(p)) {
        case 0: return (getenv("EG3D_CONV8") ? launch_conv<128, 128, 2, 4>(p, st) : launch_conv<128, 128, 2, 2>(p, st));
        case 1: return launch_conv<64, 128, 2, 2>(p, st);
        case 2: return launch_conv<32, 128, 1, 4>(p, st);
        default: return launch_conv<128, 32, 4, 1>(p, st);
    }
}

extern "C" int eg3d_conv2d_igemm_config(const eg3d_conv_params* pp) {
    if (!pp || pp->ncls < 1 || pp->ncls > 4) return EG3D_ERR_INVALID;
    return pick_config(*pp);
}
