// Low-resolution modulated convolution for gfx950: the 4^2 .. 64^2 layers of the StyleGAN2 backbone at one image per GPU
// (training/networks_stylegan2.py:34-91,417-461), forward and data gradient, in the arithmetic of conv_v2.hip (two-piece fp16 operands,
// three v_mfma_f32_32x32x16_f16 products per fp32 product, fp32 accumulation).
//
// Why a kernel of its own.  At 16 .. 4096 cells the 256-cell x 128-channel tiles of conv_v2.hip give 1 .. 64 workgroups, and the loader-split
// implicit GEMM (conv_igemm.hip) that ran these layers walked the 4608-deep contraction in 288 barrier-separated steps with two steps of loads
// in flight: 10 .. 97 us per launch at 2 .. 22 % matrix-pipe use, atomics for split-K, a zero fill in front and a finishing pass behind.
// These layers are bound by STREAMING THE WEIGHTS (9.4 MB of fp16 pieces per 512 x 512 x 3 x 3 layer against 0.02 .. 19 GFLOP), so:
//   * tile = 64 cells x 128 channels, the cells a (64 >> logw) x (1 << logw) patch: a 16 x 16 image is four tiles of 4 x 16, an 8 x 8 image one
//     tile, nothing is padded to 32 columns;
//   * the weight tile of a step (one tap of one 16-channel chunk, 8 KB) goes through an EIGHT-slot LDS ring by LDS-DMA: seven steps = 56 KB in
//     flight per workgroup, no registers, counted s_waitcnt;
//   * the A operand is the fp32 activation itself: the tile's halo of a chunk ((TR + 2) x (TW + 2) pixels x 16 channels) is DMA-ed raw into LDS
//     one chunk ahead and turned into the two fp16 piece planes there (x style x 2^e, h = rtz16, l = rne16((v - h) 2^11)) once per chunk -- no
//     operand image, no split pass, and all nine taps read the planes with a constant offset;
//   * the contraction is split over `ksplit` workgroups per tile with NO atomics: a workgroup stores its partial tile into a slab, takes a
//     ticket, and the last arriver adds the ksplit slabs in slice order and runs the fused epilogue (conv_v2_common.h: forward epilogue, or data
//     gradient + style gradient + the producing layer's activation backward).  No zero fill, no finishing pass, and the sum does not depend
//     on the arrival order (the visibility protocol is the release / ticket / acquire form of MI355X_MICROARCH.md, "Workgroup dispatch").
#include "conv_v2_common.h"
#ifndef LRV
#define LRV 0
#endif

namespace {

constexpr int LR_NB = 8;                         // weight ring slots (power of two)
constexpr int LR_BSLOT = 4 * BPLANE;             // 8 KB: [piece 2][k-octet 2][128 channels][8 x fp16]
constexpr int LR_MAXCK = 1024;

template <int RPW> struct lr_geom {
    static constexpr int CELLS = 64 * RPW;
    static constexpr int MAXSLOTS = RPW == 1 ? 144 : (RPW == 2 ? 208 : 352);   // (TR + 2) x (TW + 2) halo pixels: 4 x 34 | 6 x 18 | 10 x 10 | 18 x 6  (RPW 2: 6 x 34 ...; RPW 4: 10 x 34, 18 x 18, 34 x 10)
    static constexpr int NLD = (MAXSLOTS * 4 + 255) / 256;         // 16-byte DMA items per lane and chunk (a wave issues NLD instructions)
    static constexpr int APL = MAXSLOTS * 16;                      // one (piece, k-octet) plane of the split halo
    static constexpr int ABUF = 4 * APL;
    static constexpr int LDS_A = 0;
    static constexpr int LDS_RAW = 2 * ABUF;                       // raw fp32 halo of the NEXT chunk: [item q = 4 slot + quad][16 B]
    static constexpr int RAWBYTES = NLD * 4 * 1024;
    static constexpr int LDS_B = LDS_RAW + RAWBYTES;
    static constexpr int LDS_STY = LDS_B + LR_NB * LR_BSLOT;       // style x 2^e of this workgroup's K slice
    static constexpr int LDS_MAIN = LDS_STY + LR_MAXCK * 4;
    static constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct lr_launch { int tiles_m_max; };

template <bool FULL, int RPW>
__global__ void __launch_bounds__(256) conv_lr_kernel(const eg3d_conv_lr_params P, const int tiles_m_max) {
    using G = lr_geom<RPW>;
    constexpr int NLD = G::NLD, APL = G::APL, ABUF = G::ABUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[4];
    __shared__ unsigned s_ticket;
    const eg3d_conv_v2_params& p = P.v;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logw = P.logw, TW = 1 << logw, TR = G::CELLS >> logw;
    const int ntile_n = p.Nc / BN;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    // ---- which (class, K slice, channel tile, cell tile): cell tiles fastest, so that the workgroups streaming the same weight slice
    //      are neighbours in the XCD-contiguous order (one L2 fetches the slice once)
    int L = eg3d_xcd_remap(blockIdx.x, gridDim.x);
    const int m = L % tiles_m_max; L /= tiles_m_max;
    const int n_t = L % ntile_n; L /= ntile_n;
    const int kslice = L % ks;
    const int ci = L / ks;
    const eg3d_conv_class& cl = p.cls[ci];
    const int Ha = cl.Ha, Wa = cl.Wa, ntaps = cl.ntaps;
    const int tiles_x = (Wa + TW - 1) >> logw, tiles_y = (Ha + TR - 1) / TR;
    if (m >= p.N * tiles_y * tiles_x) return;
    const int tx = m % tiles_x, ty = (m / tiles_x) % tiles_y, n = m / (tiles_x * tiles_y);
    const int y0 = ty * TR, x0 = tx * TW, n0 = n_t * BN;
    const int tile_id = (ci * tiles_m_max + m) * ntile_n + n_t;
    const int nchunk = p.Ck / 16;
    const int c0 = (int)((int64_t)kslice * nchunk / ks), c1 = (int)((int64_t)(kslice + 1) * nchunk / ks);
    int dymin = cl.dy[0], dymax = cl.dy[0], dxmin = cl.dx[0], dxmax = cl.dx[0];
    for (int t = 1; t < ntaps; ++t) {
        dymin = min(dymin, cl.dy[t]); dymax = max(dymax, cl.dy[t]);
        dxmin = min(dxmin, cl.dx[t]); dxmax = max(dxmax, cl.dx[t]);
    }
    const int HC = TW + dxmax - dxmin, HR = TR + dymax - dymin, hslots = HR * HC;      // halo of the tile (<= MAXSLOTS: host check)
    // per-tap constants live in LANE t of two registers and are fetched with v_readlane: indexing the class table by a loop variable would be a
    // scalar load from the kernel-argument segment (host memory: ~0.35 us each) in every step -- measured: 288 steps took 125 us whatever else ran
    const int tl = lane < ntaps ? lane : 0;
    const int tap_off_v = (cl.dy[tl] * HC + cl.dx[tl]) * 16;
    const int tap_w_v = cl.wtap[tl];
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    constexpr unsigned OOB = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * p.Hi * p.Wi * P.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);

    // ---- styles of this slice and max|style| (every workgroup derives the same range scale itself: a few KB out of L2) -----------------
    // chunk walk: logical position j = 0 .. nck-1 -> chunk c0 + (j + rot) % nck
    const int nck = c1 - c0;
    const int rot = P.rotate ? (int)(((int64_t)m * nck) / max(1, tiles_m_max)) % nck : 0;
    auto phys = [&](int j) { const int q = j + rot; return c0 + (q >= nck ? q - nck : q); };
    const int nsty = (c1 - c0) * 16;
    float sreg[LR_MAXCK / 256];
    float smaxv = 0.f;
    if (P.in_scale != nullptr) {
        for (int i = tid; i < p.N * p.Ck; i += 256) smaxv = fmaxf(smaxv, fabsf(P.in_scale[i]));
#pragma unroll
        for (int j = 0; j < LR_MAXCK / 256; ++j) {
            const int k = tid + 256 * j;
            sreg[j] = k < nsty ? P.in_scale[(int64_t)n * p.Ck + c0 * 16 + k] : 0.f;
        }
    } else {
        smaxv = 1.f;
#pragma unroll
        for (int j = 0; j < LR_MAXCK / 256; ++j) sreg[j] = 1.f;
    }
    const float x_amax = *P.x_amax * P.amax_mul;

    // ---- A loader: the wave's DMA instructions i = 0 .. NLD-1 cover the items q = (wave + 4 i) 64 + lane; item = (halo slot q >> 2, channel
    //      quad q & 3), 16 bytes = 4 channels of one pixel; the same wave converts the items it fetched (its own vmcnt covers them)
    unsigned a_goff[NLD];
    int a_slot[NLD];
    const int quad = lane & 3;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;
        const int slot = q >> 2;
        const int hy = slot / HC, hx = slot - hy * HC;
        const int y = y0 + dymin + hy, x = x0 + dxmin + hx;
        const bool ok = slot < hslots && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
        a_goff[i] = ok ? (unsigned)((((n * p.Hi + y) * p.Wi + x) * P.ldx + quad * 4) * 4) : OOB;
        a_slot[i] = slot < hslots ? slot : -1;
    }
    auto issue_A = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            glds16(xrs, lds0 + G::LDS_RAW + (wave + 4 * i) * 1024, a_goff[i] == OOB ? OOB : a_goff[i] + (unsigned)(chunk * 64));
    };
    auto issue_B = [&](int chunk, int tap, int slot, bool live) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = wave * 2 + e, plane = idx >> 1, half = idx & 1;
            const unsigned v = (unsigned)(((((__builtin_amdgcn_readlane(tap_w_v, tap) * nchunk + chunk) * 4 + plane) * p.Nc) + n0 + half * 64 + lane) * 16);
            glds16(wrs, lds0 + G::LDS_B + slot * LR_BSLOT + plane * BPLANE + half * 1024, (!live || (!FULL && plane >= 2)) ? OOB : v);
        }
    };
    auto convert_A = [&](int j) {           // raw halo of walk position j -> the two fp16 piece planes of buffer j & 1
        const int chunk = phys(j);
        // (ext-vector typed LDS reads: a HIP float4 load here makes the compiler's LDS-DMA hazard tracking put s_waitcnt vmcnt(0) in front of
        //  it, i.e. drain the whole weight ring once per chunk)
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(smem + G::LDS_STY + ((chunk - c0) * 16 + quad * 4) * 4);
        char* dst = smem + G::LDS_A + (j & 1) * ABUF + (quad >> 1) * APL + (quad & 1) * 8;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (a_slot[i] < 0) continue;
            const f32x4 r = *reinterpret_cast<const f32x4*>(smem + G::LDS_RAW + ((wave + 4 * i) * 64 + lane) * 16);
            const float v[4] = {r[0] * s4[0], r[1] * s4[1], r[2] * s4[2], r[3] * s4[3]};
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const fp16x2_t hh = __builtin_amdgcn_cvt_pkrtz(v[2 * e], v[2 * e + 1]);
                h[2 * e] = (_Float16)hh[0]; h[2 * e + 1] = (_Float16)hh[1];
                l[2 * e] = (_Float16)__builtin_amdgcn_fmed3f((v[2 * e] - (float)hh[0]) * 2048.f, -65504.f, 65504.f);
                l[2 * e + 1] = (_Float16)__builtin_amdgcn_fmed3f((v[2 * e + 1] - (float)hh[1]) * 2048.f, -65504.f, 65504.f);
            }
            *reinterpret_cast<f16x4*>(dst + a_slot[i] * 16) = h;
            if constexpr (FULL) *reinterpret_cast<f16x4*>(dst + 2 * APL + a_slot[i] * 16) = l;
        }
    };

    // ---- prologue: A(c0) and the first NB - 1 weight steps go out before anything waits ---------------------------------------------------
    const int S = (c1 - c0) * ntaps;
    issue_A(phys(0));
    int bc = 0, bt = 0, bstep = 0;                 // next weight step to issue: (walk position, tap), ring slot bstep & (NB - 1)
    auto issue_next_B = [&]() {
        issue_B(phys(bc < nck ? bc : nck - 1), bt, bstep & (LR_NB - 1), bstep < S);
        ++bstep;
        if (++bt == ntaps) { bt = 0; ++bc; }
    };
#pragma unroll
    for (int s = 0; s < LR_NB - 1; ++s) issue_next_B();

#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) smaxv = fmaxf(smaxv, __shfl_xor(smaxv, o));
    if (lane == 0) red[wave] = smaxv;
    __syncthreads();
    const float smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mul = range_mul(x_amax * smax);
    const float out_mul = 1.f / (mul * *p.w_scale);                         // exact powers of two
#pragma unroll
    for (int j = 0; j < LR_MAXCK / 256; ++j) {
        const int k = tid + 256 * j;
        if (k < nsty) reinterpret_cast<float*>(smem + G::LDS_STY)[k] = sreg[j] * mul;
    }
    __syncthreads();
    wait_vm_n<2 * (LR_NB - 1)>();                  // A(c0) landed (the weight steps behind it stay in flight)
    convert_A(0);

    f32x16 acc[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned a_lane[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int lin = (wm * RPW + i) * 32 + (lane & 31);
        const int cy = lin >> logw, cx = lin & (TW - 1);
        a_lane[i] = (unsigned)(((cy - dymin) * HC + cx - dxmin) * 16 + (lane >> 5) * APL);
    }
    const unsigned b_lane = (unsigned)((wn * 64 + (lane & 31)) * 16 + (lane >> 5) * BPLANE);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    // ---- main loop: one flat loop over the steps (tap fastest); a nested chunk / tap loop made the compiler keep the accumulators in VGPRs
    //      across the outer loop and copy all 32 to and from the AGPRs once per chunk
    int tap = 0, chunk = 0, a_age = 1000;          // chunk: walk position 0 .. nck-1; a_age: steps since the last A issue
    for (int step = 0; step < S; ++step, ++a_age) {
        const bool more = chunk + 1 < nck;
        // B(step) must have landed.  Younger operations of this wave: the weight steps step+1 .. step+NB-2 (2 each) and, for NB - 1 steps
        // after an A issue, the NLD instructions of that halo
        if (a_age <= LR_NB - 1) wait_vm_n<2 * (LR_NB - 2) + NLD>();
        else wait_vm_n<2 * (LR_NB - 2)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the piece planes written at the end of the previous chunk
#if LRV != 3
        __builtin_amdgcn_s_barrier();
#endif
#if LRV != 1
        issue_next_B();
#endif
        if (tap == 0 && more) { issue_A(phys(chunk + 1)); a_age = 0; }
        const unsigned abase = G::LDS_A + (chunk & 1) * ABUF + (unsigned)__builtin_amdgcn_readlane(tap_off_v, tap);
        const unsigned bbase = G::LDS_B + (step & (LR_NB - 1)) * LR_BSLOT + b_lane;
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
#if LRV != 2 && LRV != 4
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(smem + abase + a_lane[i]);
            if constexpr (FULL) {
                const f16x8 al = *reinterpret_cast<const f16x8*>(smem + abase + a_lane[i] + 2 * APL);
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
#elif LRV == 2
        acc[0][0][0] += (float)bh[0][0] + (float)bl[1][3] + (float)bg[0][1];      // keeps the B reads
#endif
        if (tap == ntaps - 1) {
            if (more) {
                // the halo of chunk + 1 was issued at tap 0; behind it: the weight steps of taps 1 .. ntaps-1
                if (ntaps >= 9) { /* 14 younger operations by the top of this step, whose wait (<= 12 outstanding: a_age = 8) covered it */ }
                else if (ntaps >= 4) wait_vm_n<6>();
                else if (ntaps >= 2) wait_vm_n<2>();
                else wait_vm_n<0>();
                convert_A(chunk + 1);
            }
            tap = 0; ++chunk;
        } else {
            ++tap;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (ks > 1) {
        // ---- split-K without atomics: slab store -> release -> ticket; the last arriver sums the slabs in slice order ---------------------
        constexpr int NE = RPW * 2 * 4;                       // float4 elements per thread
        float4* slab = reinterpret_cast<float4*>(P.slabs) + ((int64_t)tile_id * ks + kslice) * NE * 256 + tid;
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    slab[((i * 2 + j) * 4 + r4) * 256] = make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_ticket = __hip_atomic_fetch_add(P.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_ticket != (unsigned)(ks - 1)) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(P.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero again for the next launch
        }
        __syncthreads();
        const float4* s0 = reinterpret_cast<const float4*>(P.slabs) + (int64_t)tile_id * ks * NE * 256 + tid;
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int s = 0; s < ks; ++s) {
            float4 v[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) v[e] = s0[((int64_t)s * NE + e) * 256];
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float4 t = v[(i * 2 + j) * 4 + r4];
                        acc[i][j][4 * r4] += t.x; acc[i][j][4 * r4 + 1] += t.y; acc[i][j][4 * r4 + 2] += t.z; acc[i][j][4 * r4 + 3] += t.w;
                    }
        }
    }
    v2_epilogue<false, RPW, true>(p, acc, Ha, Wa, cl.out_py, cl.out_px, n, y0, x0, n0, smem, out_mul, logw);
}

std::atomic<uint64_t> g_lr_attr[6];

inline int lr_tiles_m_max(const eg3d_conv_lr_params& P, int rpw) {
    const int TW = 1 << P.logw, TR = (64 * rpw) >> P.logw;
    int mx = 0;
    for (int c = 0; c < P.v.ncls; ++c)
        mx = std::max(mx, P.v.N * eg3d_cdiv(P.v.cls[c].Ha, TR) * eg3d_cdiv(P.v.cls[c].Wa, TW));
    return mx;
}

template <bool FULL, int RPW>
int launch_lr(const eg3d_conv_lr_params& P, hipStream_t st, int slot) {
    auto kern = conv_lr_kernel<FULL, RPW>;
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lr_geom<RPW>::LDS_BYTES, g_lr_attr[slot])) return e;
    const int tm = lr_tiles_m_max(P, RPW);
    const int ks = P.v.ksplit > 1 ? P.v.ksplit : 1;
    const int64_t total = (int64_t)tm * (P.v.Nc / BN) * ks * P.v.ncls;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lr_geom<RPW>::LDS_BYTES, st, P, tm);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

}  // namespace

extern "C" int eg3d_conv2d_lr_supported(const eg3d_conv_lr_params* pp) {
    if (!pp) return 0;
    const eg3d_conv_v2_params& p = pp->v;
    if (p.N <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ck < 16 || (p.Ck & 15) || p.Ck > LR_MAXCK || p.Nc < BN || (p.Nc % BN) || (p.ldo & 3)) return 0;
    if (p.in_stride != 1 || p.out_stride < 1 || p.ncls < 1 || p.ncls > 4) return 0;
    if (pp->logw < 2 || pp->logw > 5 || pp->ldx < p.Ck || (pp->ldx & 3)) return 0;
    if (p.products != 0 && p.products != 1 && p.products != 3) return 0;
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_FWD && p.epi != EG3D_EPI_BWD && p.epi != EG3D_EPI_BWD_ACT) return 0;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    if (ks > 16 || ks > p.Ck / 16) return 0;
    if (p.epi == EG3D_EPI_FWD && !eg3d_act_is_pwl(p.act)) return 0;
    if (p.epi == EG3D_EPI_BWD_ACT) {
        const eg3d_act_bwd& ab = p.act_bwd;
        if (ab.act != EG3D_ACT_LINEAR && ab.act != EG3D_ACT_LRELU) return 0;
        if (!(ab.gain > 0.f) || (ab.noise != nullptr && ab.noise_strength == nullptr)) return 0;
    }
    const int rpw = p.patch_rows == 4 ? 4 : (p.patch_rows == 2 ? 2 : 1);
    if (p.patch_rows != 0 && p.patch_rows != 1 && p.patch_rows != 2 && p.patch_rows != 4) return 0;
    if (rpw == 4 && pp->logw < 3) return 0;
    const int TW = 1 << pp->logw, TR = (64 * rpw) >> pp->logw;
    const int maxslots = rpw == 1 ? lr_geom<1>::MAXSLOTS : (rpw == 2 ? lr_geom<2>::MAXSLOTS : lr_geom<4>::MAXSLOTS);
    for (int c = 0; c < p.ncls; ++c) {
        const eg3d_conv_class& k = p.cls[c];
        if (k.ntaps != 9 && k.ntaps != 4 && k.ntaps != 2 && k.ntaps != 1) return 0;
        int ymin = k.dy[0], ymax = k.dy[0], xmin = k.dx[0], xmax = k.dx[0];
        for (int t = 1; t < k.ntaps; ++t) { ymin = std::min(ymin, k.dy[t]); ymax = std::max(ymax, k.dy[t]); xmin = std::min(xmin, k.dx[t]); xmax = std::max(xmax, k.dx[t]); }
        if (ymax - ymin > 2 || xmax - xmin > 2) return 0;
        if ((TR + ymax - ymin) * (TW + xmax - xmin) > maxslots) return 0;
        for (int t = 0; t < k.ntaps; ++t) if (k.wtap[t] < 0 || k.wtap[t] >= p.wtaps) return 0;
    }
    if ((int64_t)p.N * p.Hi * p.Wi * pp->ldx * 4 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.wtaps * (p.Ck / 16) * 4 * p.Nc * 16 > 0x7fffffe0ll) return 0;
    if ((int64_t)p.N * p.Ho * p.Wo * p.ldo > INT32_MAX) return 0;
    return 1;
}

extern "C" int eg3d_conv2d_lr_workspace(const eg3d_conv_lr_params* pp, int64_t* slab_bytes, int64_t* ticket_words) {
    if (!pp || !slab_bytes || !ticket_words) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_lr_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const int ks = pp->v.ksplit > 1 ? pp->v.ksplit : 1;
    if (ks == 1) { *slab_bytes = 0; *ticket_words = 0; return EG3D_OK; }
    const int rpw = pp->v.patch_rows == 4 ? 4 : (pp->v.patch_rows == 2 ? 2 : 1);
    const int64_t tiles = (int64_t)lr_tiles_m_max(*pp, rpw) * (pp->v.Nc / BN) * pp->v.ncls;
    *slab_bytes = tiles * ks * 64 * rpw * BN * 4;
    *ticket_words = tiles;
    return EG3D_OK;
}

extern "C" int eg3d_conv2d_lr(const eg3d_conv_lr_params* pp, void* stream) {
    if (!pp || !pp->v.a || !pp->v.w || !pp->v.out || !pp->v.w_scale || !pp->x_amax) return EG3D_ERR_INVALID;
    if (!eg3d_conv2d_lr_supported(pp)) return EG3D_ERR_UNSUPPORTED;
    const eg3d_conv_v2_params& p = pp->v;
    if (p.ksplit > 1 && (!pp->slabs || !pp->tickets)) return EG3D_ERR_INVALID;
    if (p.epi == EG3D_EPI_BWD_ACT && !p.xin) return EG3D_ERR_INVALID;
    const void* ptrs[] = {p.a, p.out, p.addend, p.xin, p.out_scale, p.bias, p.act_bwd.d, p.act_bwd.bias, pp->slabs};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.epi == EG3D_EPI_FWD && p.noise && !p.noise_strength) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const bool one = p.products == 1;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND_V2(det, p); EG3D_DET_COMMIT(det);
    int rc;
    if (p.patch_rows == 4) rc = one ? launch_lr<false, 4>(*pp, st, 5) : launch_lr<true, 4>(*pp, st, 4);
    else if (p.patch_rows == 2) rc = one ? launch_lr<false, 2>(*pp, st, 3) : launch_lr<true, 2>(*pp, st, 2);
    else rc = one ? launch_lr<false, 1>(*pp, st, 1) : launch_lr<true, 1>(*pp, st, 0);
    EG3D_DET_END(det);
    return rc;
}
