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
#include "conv_v2_common.h"

namespace {
// ---- issue schedule of conv_v2_kernel: what ONE wave issues at a step = (tap, is this the last chunk), in issue order ----------------------------------
// prologue: all NPARTS A parts of the first chunk, B(step 0), B(step 1);   step (tap, last): n_a A parts of the NEXT chunk, then the two B operations of step + 2.
// The kernel's issue loops AND its vmcnt immediates are both taken from these functions, and tools/rootcause/isa_protocol.py compares them with the compiled code.
template <int NTAPS, int NPARTS>
struct v2_sched {
    static constexpr int APT = (NPARTS + NTAPS - 1) / NTAPS;          // A parts a wave issues per step ...
    static constexpr int NA_TAPS = NPARTS / APT;                      // ... during the first NA_TAPS taps of a chunk
    static_assert(NPARTS % APT == 0 && NA_TAPS <= NTAPS, "A parts per step");
    static constexpr int n_a(int tap, bool last) { return (!last && tap < NA_TAPS) ? APT : 0; }
    // B(step + 2) is issued unless it lies past the last chunk (one-tap classes decide that at run time and therefore always wait for everything)
    static constexpr bool b_static(int tap, bool last) { return !last || tap + 2 < NTAPS; }
    static constexpr int n_b(int tap, bool last) { return b_static(tap, last) ? 2 : 0; }
    // LDS-DMA operations that may still be in flight at the boundary in front of step (tap, last): what the PREVIOUS step issued after B(this step) -- its A parts
    // (they precede its B operations) and its B operations -- except at tap 0, where the A parts are this chunk's own tile and must have landed as well.
    // (the step before tap 0 is the last tap of a chunk that is not the last one -- or the prologue, which ends with the same two B operations)
    static constexpr int allow(int tap, bool last) {
        if (NTAPS == 1) return 0;
        const int ptap = tap >= 1 ? tap - 1 : NTAPS - 1;
        const bool plast = tap >= 1 ? last : false;
        return n_b(ptap, plast) + (tap == 0 ? 0 : n_a(ptap, plast));
    }
    static constexpr int total_a() { int t = 0; for (int i = 0; i < NTAPS; ++i) t += n_a(i, false); return t; }
    static_assert(total_a() == NPARTS, "every A part of the next chunk is issued exactly once per chunk");
    static_assert(allow(NTAPS - 1, true) == 0 || NTAPS == 1, "the last step waits for everything");
};

// FULL: three products per fp32 product; !FULL: high pieces only (EG3D_PREC_F16X1); ATOMIC: split-K; RPW: patch rows per wave -- 4 = the 8 x 32
// patch (256 cells), 2 = a 4 x 32 patch (128 cells x 128 channels per workgroup, 12 MFMAs per wave and step): twice the workgroups for the
// layers whose 8-row grids leave CUs idle (128^2 x 256: 128 -> 256), with the fused epilogues intact (split-K needs a zero fill + a finishing pass);
// 1 = a 2 x 32 patch (64 cells: 64^2 x 512 -> 256 workgroups)
// KH = 2 (4 x 32 patches whose grid gives every CU ONE workgroup: 128^2 x 256, 256^2 x 128): the workgroup has eight waves -- two independent halves,
// each the four-wave kernel on one half of the contraction with its own LDS image (A double buffer + weight ring) -- so that every SIMD holds two waves and
// one half's matrix instructions issue under the other's barrier / LDS-read latencies (with one wave per SIMD nothing does: MfmaUtil 33 % against 57 % for
// the 8-row launches that have two workgroups per CU).  The halves meet once, in LDS, after their last step; the upper half then ends (a barrier does not
// wait for waves that have ended) and the lower one runs the fused epilogue unchanged.
template <int NTAPS, bool FULL = true, bool ATOMIC = false, int RPW = 4, bool RGB = false, int KH = 1>          // RGB: + the 1x1 head of the forward epilogue (eg3d_conv_v2_params::rgb_out)
__global__ void __launch_bounds__(256 * KH, KH == 1 ? 2 : 1) conv_v2_kernel(const eg3d_conv_v2_params p, const int cls_base) {
    constexpr int PHK = 2 * RPW;                          // patch rows of this instantiation
    constexpr int NPARTS = ((PHK + 2) * (PW + 2) + 63) / 64;      // 64-slot wave-instructions per A plane: halo <= (PHK + 2) x 34 slots (6 | 4 | 3)
    using sched = v2_sched<NTAPS, NPARTS>;                // issue counts per step and the vmcnt immediates that follow from them
    extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    // (KH = 1 must compile to the instructions it had before KH existed: written with `wave = wave_all & 3` for both forms, every instantiation of this
    //  kernel came out 10 - 150 instructions different and the deterministic build stopped being bit-identical run to run -- sporadic 1e-7 .. 1e-5
    //  differences, the signature of the counted-vmcnt / LDS-DMA protocol being disturbed -- see the Makefile note on this kernel's other sensitivity)
    int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    int khalf = 0;
    if constexpr (KH == 2) { khalf = wave_ >> 2; wave_ &= 3; }
    const int wave = wave_;
    char* const smem = KH == 2 ? smem_dyn + khalf * LDS_MAIN : smem_dyn;          // this half's LDS image
    const int wm = wave >> 1, wn = wave & 1;
    const eg3d_conv_class& cl = p.cls[cls_base + blockIdx.z];
    const int Ha = cl.Ha, Wa = cl.Wa;
    const int tiles_x = (Wa + PW - 1) / PW, tiles_y = (Ha + PHK - 1) / PHK, ntile_n = p.Nc / BN;
    const int ntile = p.N * tiles_y * tiles_x * ntile_n;
    int bid = blockIdx.x;
    if (bid >= ntile) return;
    bid = eg3d_xcd_remap(bid, ntile);
    const int n_t = bid % ntile_n; bid /= ntile_n;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int n = bid / tiles_y;
    const int y0 = ty * PHK, x0 = tx * PW, n0 = n_t * BN;
    const int nchunk = p.Ck / 16;
    // split-K (EG3D_EPI_ATOMIC): blockIdx.y owns the 16-channel chunks [c0, c1) of the contraction and adds its partial tile to `out`
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    int c0 = (int)((int64_t)blockIdx.y * nchunk / ks), c1 = (int)((int64_t)(blockIdx.y + 1) * nchunk / ks);
    if constexpr (KH == 2) { c0 = khalf * (nchunk / 2); c1 = c0 + nchunk / 2; }          // (nchunk even, host check: both halves make the same number of steps, i.e. meet at the same barriers)
    const int planeA = p.Hi * p.Wi * 16;                   // bytes of one (piece, k-octet) plane of the A image
    // tap extent of this class -> halo geometry
    int dymin = cl.dy[0], dymax = cl.dy[0], dxmin = cl.dx[0], dxmax = cl.dx[0];
#pragma unroll
    for (int t = 1; t < NTAPS; ++t) {
        dymin = min(dymin, cl.dy[t]); dymax = max(dymax, cl.dy[t]);
        dxmin = min(dxmin, cl.dx[t]); dxmax = max(dxmax, cl.dx[t]);
    }
    constexpr int hw = PW + 2;                                        // LDS row pitch of the halo: always 34 pixels (columns past the
    const int hh = PHK + dymax - dymin;                               // tap extent are loaded but never read); <= PHK + 2 rows (host check)
    const int hslots = hw * hh;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;                  // LDS byte address of the dynamic array

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)((int64_t)p.N * 2 * (p.Ck / 8) * planeA), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((int64_t)p.wtaps * nchunk * 4 * p.Nc * 16), 0x00020000);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- A loader: wave w issues the wave-instructions j = w + 4 i (i = 0..5) of a chunk: plane j / 6, 64-slot part j % 6 ------------
    unsigned a_pix[NPARTS];
    int a_plane[NPARTS], a_part[NPARTS];
#pragma unroll
    for (int i = 0; i < NPARTS; ++i) {
        const int j = wave + 4 * i;
        a_plane[i] = j / NPARTS; a_part[i] = j % NPARTS;
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
    auto issue_B = [&](int chunk, int wtap, int slot) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = wave * 2 + e, plane = idx >> 1, half = idx & 1;
            const unsigned v = (unsigned)(((((wtap * nchunk + chunk) * 4 + plane) * p.Nc) + n0 + half * 64 + lane) * 16);
            glds16(wrs, lds0 + LDS_B + slot * BSLOT + plane * BPLANE + half * 1024, (!FULL && plane >= 2) ? OOB : v);
        }
    };

    f32x16 acc[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned a_lane = (unsigned)(((wm * RPW) * hw + (lane & 31)) * 16 + (lane >> 5) * APLANE);          // (+ the tap's offset from the halo origin: tap_off)
    const unsigned b_lane = (unsigned)((wn * 64 + (lane & 31)) * 16 + (lane >> 5) * BPLANE);
    const f16x2 k2m11 = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};

    // the class's tap tables in scalar registers: read through `cl` inside the loop they are re-fetched from the kernel-argument segment after every boundary
    // (the boundary clobbers memory), an s_load + s_waitcnt lgkmcnt(0) in front of every step's first matrix instruction
    // (one register per tap: LDS byte offset of the tap from the halo origin, < 2^16, | weight-tap index << 16)
    unsigned tap_tab[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)          // offsets relative to the halo origin (dymin, dxmin): non-negative, <= (2 * 34 + 2) * 16
        tap_tab[t] = (unsigned)__builtin_amdgcn_readfirstlane((((cl.dy[t] - dymin) * hw + (cl.dx[t] - dxmin)) * 16) | (cl.wtap[t] << 16));
    auto tap_w = [&](int t) { return (int)(tap_tab[t] >> 16); };
    auto tap_off = [&](int t) { return tap_tab[t] & 0xffffu; };

    // ---- prologue: A(c0), B(step 0), B(step 1) -------------------------------------------------------------------------------------------
    const int S = (c1 - c0) * NTAPS;
#pragma unroll
    for (int i = 0; i < NPARTS; ++i) issue_A(c0, i);
    issue_B(c0, tap_w(0), 0);
    if (S > 1) issue_B(NTAPS > 1 ? c0 : c0 + 1, tap_w(NTAPS > 1 ? 1 : 0), 1);
    else { issue_B(c0, tap_w(0), 1); }               // keeps the wait accounting uniform (never read)

    int step = 0;
    auto run_chunk = [&](const int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        static_for<0, NTAPS>([&](auto tap_tag) {
            constexpr int tap = decltype(tap_tag)::value;
            // B(step) -- and at tap 0 all of A(chunk) -- have landed, everybody's LDS reads of the previous step have returned
            step_sync<sched::allow(tap, LAST)>();
#pragma unroll
            for (int e = 0; e < sched::n_a(tap, LAST); ++e) issue_A(chunk + 1, tap * sched::APT + e);
            {
                constexpr int t2 = (tap + 2) % NTAPS, dc = (tap + 2) / NTAPS;
                if (NTAPS == 1 ? chunk + dc < c1 : sched::b_static(tap, LAST)) issue_B(chunk + dc, tap_w(t2), (step + 2) % 3);
            }
            const unsigned abase = LDS_A + (chunk & 1) * ABUF + a_lane + tap_off(tap);
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
            for (int i = 0; i < RPW; ++i) {
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
    step_sync<0>();                      // every LDS-DMA and LDS read of the main loop is over: the exchange / the epilogue re-use the dynamic LDS
    if constexpr (KH == 2) {
        // the upper half's partial tile goes to the lower one through the upper half's (now idle) LDS image: [i][j][r][thread], lane-contiguous
        float* xch = reinterpret_cast<float*>(smem_dyn + LDS_MAIN) + (wave * 64 + lane);
        if (khalf == 1) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((i * 2 + j) * 16 + r) * 256] = acc[i][j][r];
        }
        __syncthreads();
        // The upper half ends here; the lower one goes on through the barriers of the epilogue.  That is defined behaviour of the INSTRUCTION, not of the HIP model:
        // S_BARRIER waits for the waves of the workgroup that have not terminated ("if some waves in the threadgroup have already terminated, this waits on only the
        // surviving waves", CDNA ISA, SOPP S_BARRIER) -- the epilogue and eg3d_commit_amax_block are told the surviving wave count (nwaves = 4) for everything that is
        // indexed by wave.  tests/test_gpu_ops.py::test_conv_v2_k_halves_equal_the_four_wave_form holds it to the four-wave form and to launch-to-launch bit-identity.
        if (khalf == 1) return;
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[((i * 2 + j) * 16 + r) * 256];
    }

    if constexpr (KH == 2) v2_epilogue<ATOMIC, RPW, false, RGB>(p, acc, Ha, Wa, cl.out_py, cl.out_px, n, y0, x0, n0, smem, 1.f / (*p.a_scale * *p.w_scale), 5, 4);
    else v2_epilogue<ATOMIC, RPW, false, RGB>(p, acc, Ha, Wa, cl.out_py, cl.out_px, n, y0, x0, n0, smem, 1.f / (*p.a_scale * *p.w_scale));
}

// ---- operand preparation (split8 / range_mul: conv_v2_common.h) ------------------------------------------------------------
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
    // blockIdx.y owns a group of channel octets: small images (128^2 and below: 64 blocks of pixels) would otherwise leave most CUs idle
    const int og = (noct + gridDim.y - 1) / gridDim.y;
    const int ko_end = min(noct, (int)(blockIdx.y + 1) * og);
    for (int ko = blockIdx.y * og; ko < ko_end; ++ko) {
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

// Same pass through LDS: the loads run along the channel axis (a wave-instruction = whole 512-byte pixel rows), the stores along the
// pixel axis (1 KB runs of one (piece, octet) plane).  Block = 64 pixels x CH channels.  The thread-per-pixel form above reads 32 bytes
// per lane from 64 different rows per instruction: 3.6 TB/s at 512^2 x 128; and one block per 256 pixels leaves a 64^2 image on 16 CUs.
template <int CH>
__global__ void __launch_bounds__(256) split_act_lds_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* x_amax, const float* s_amax,
                                                            f16x8* __restrict__ out, float* scale_out, int N, int HW, int C, int ldx) {
    constexpr int TP = 64, PITCH = CH + 4, F4 = CH / 4, PPP = 256 / F4;          // pixels per load pass
    __shared__ __attribute__((aligned(16))) float tile[TP * PITCH];
    __shared__ float red[4];
    const int64_t npix = (int64_t)N * HW;
    const int64_t pix0 = (int64_t)blockIdx.x * TP;
    const int c0 = blockIdx.y * CH;
    const int f4 = threadIdx.x % F4, pl = threadIdx.x / F4;
    // The tile's loads go out FIRST; max|in_scale| (every block derives the same value from a few KB), the range scalar and this thread's style values are
    // requested while they are in flight and everything meets at ONE barrier.  In front of the tile loads the reduction was a dependent memory round trip plus
    // a barrier of its own per block, and the styles a third round trip after the barrier (2 - 2.5 us each on a 6 - 14 us launch: DESIGN.md 3.1a).
    float4 tv[TP / PPP];
#pragma unroll
    for (int pass = 0; pass < TP / PPP; ++pass) {
        const int pq = pl + pass * PPP;
        tv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pix0 + pq < npix) tv[pass] = *reinterpret_cast<const float4*>(x + (pix0 + pq) * ldx + c0 + f4 * 4);
    }
    const int p = threadIdx.x & 63, og = threadIdx.x >> 6;                        // phase 2: pixel, group of CH / 32 octets
    const int64_t pix = pix0 + p;
    const bool live = pix < npix;
    const int n = live ? (int)(pix / HW) : 0;
    const int64_t pp = pix - (int64_t)n * HW;
    const int noct = C / 8;
    float4 sv[CH / 32][2];
#pragma unroll
    for (int k = 0; k < CH / 32; ++k) {
        sv[k][0] = sv[k][1] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (s != nullptr) {
            const float* sr = s + (int64_t)n * C + c0 + (og * (CH / 32) + k) * 8;
            sv[k][0] = *reinterpret_cast<const float4*>(sr); sv[k][1] = *reinterpret_cast<const float4*>(sr + 4);
        }
    }
    float smax = 1.f;
    const bool own_max = s_amax == nullptr && s != nullptr;
    if (s_amax != nullptr) smax = *s_amax;
    if (own_max) {
        float m = 0.f;
        for (int i = threadIdx.x; i < N * C; i += 256) m = fmaxf(m, fabsf(s[i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    }
    const float xa = *x_amax;
#pragma unroll
    for (int pass = 0; pass < TP / PPP; ++pass) *reinterpret_cast<float4*>(tile + (pl + pass * PPP) * PITCH + f4 * 4) = tv[pass];
    __syncthreads();
    if (own_max) smax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mul = range_mul(xa * smax);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *scale_out = mul;
    if (!live) return;
#pragma unroll
    for (int k = 0; k < CH / 32; ++k) {
        const int kl = og * (CH / 32) + k;                                        // octet inside the chunk
        float v[8];
        const float4 v0 = *reinterpret_cast<const float4*>(tile + p * PITCH + kl * 8), v1 = *reinterpret_cast<const float4*>(tile + p * PITCH + kl * 8 + 4);
        v[0] = v0.x * sv[k][0].x; v[1] = v0.y * sv[k][0].y; v[2] = v0.z * sv[k][0].z; v[3] = v0.w * sv[k][0].w;
        v[4] = v1.x * sv[k][1].x; v[5] = v1.y * sv[k][1].y; v[6] = v1.z * sv[k][1].z; v[7] = v1.w * sv[k][1].w;
        f16x8 h, l;
        split8(v, mul, h, l, 2048.f);
        const int ko = c0 / 8 + kl;
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

// batched forms of the two kernels above for the pivotal-tuning phase, where every weight image is stale once per step (20 launches of ~5 us)
__global__ void __launch_bounds__(256) absmax_batched_kernel(const eg3d_split_w_batch b) {
    const eg3d_split_w_item& it = b.items[blockIdx.y];
    const int64_t n = (int64_t)it.O * it.w_row;                  // dense packed matrix (w_row = T * I)
    float m = 0.f;
    const float4* x4 = reinterpret_cast<const float4*>(it.w);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += (int64_t)gridDim.x * 256) {
        const float4 v = x4[i];
        m = fmaxf(m, fmaxf(fmaxf(finite_abs(v.x), finite_abs(v.y)), fmaxf(finite_abs(v.z), finite_abs(v.w))));
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f && m < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(it.amax), __float_as_uint(m));
    }
}
__global__ void __launch_bounds__(256) split_w_batched_kernel(const eg3d_split_w_batch b) {
    const eg3d_split_w_item& it = b.items[blockIdx.y];
    const float mul = range_mul(*it.amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) *it.scale_out = mul;
    const int O = it.O, I = it.I, noct = I / 8;
    f16x8* out = reinterpret_cast<f16x8*>(it.image);
    const int64_t tot = (int64_t)O * it.T * noct;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tot; t += (int64_t)gridDim.x * 256) {
        const int o = (int)(t % O);
        const int64_t r = t / O;
        const int ko = (int)(r % noct), tap = (int)(r / noct);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = it.w[(int64_t)o * it.w_row + (int64_t)tap * I + ko * 8 + q];
        f16x8 h, l;
        split8(v, mul, h, l, 1.f);
        const int chunk = ko >> 1, koct = ko & 1;
        out[((((int64_t)tap * (I / 16) + chunk) * 2 + 0) * 2 + koct) * O + o] = h;
        out[((((int64_t)tap * (I / 16) + chunk) * 2 + 1) * 2 + koct) * O + o] = l;
    }
}

std::atomic<uint64_t> g_attr[15];

template <int NTAPS, bool FULL = true, bool ATOMIC = false, int RPW = 4, bool RGB = false, int KH = 1>
int launch_v2(const eg3d_conv_v2_params& p, int cls_base, int ncls, int max_tiles, hipStream_t st, int slot) {
    auto kern = conv_v2_kernel<NTAPS, FULL, ATOMIC, RPW, RGB, KH>;
    constexpr int LDS_K = KH == 2 ? 2 * LDS_MAIN : LDS_BYTES;          // (KH = 2: two main-loop images; the epilogue re-uses the lower one, the exchange the upper one: 64 KB <= LDS_MAIN)
    static_assert(KH == 1 || (LDS_EPI <= LDS_MAIN && RPW * 2 * 16 * 256 * 4 <= LDS_MAIN), "KH = 2 LDS re-use");
    if (int e = eg3d_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LDS_K, g_attr[slot])) return e;
    hipLaunchKernelGGL(kern, dim3(max_tiles, (KH == 1 && p.ksplit > 1) ? p.ksplit : 1, ncls), dim3(256 * KH), LDS_K, st, p, cls_base);
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
    if (p.epi != EG3D_EPI_STORE && p.epi != EG3D_EPI_FWD && p.epi != EG3D_EPI_BWD && p.epi != EG3D_EPI_BWD_ACT && p.epi != EG3D_EPI_ATOMIC) return 0;
    if (p.epi == EG3D_EPI_ATOMIC)                         // the split-K instantiations exist for the 3x3 classes
        for (int c = 0; c < p.ncls; ++c) if (p.cls[c].ntaps != 9) return 0;
    // ksplit == 2 with a fused (non-atomic) epilogue and 4-row patches: the contraction is split over the two four-wave halves of an eight-wave
    // workgroup (KH = 2 of conv_v2_kernel) -- no partial tiles leave the workgroup
    const bool khalves = p.ksplit == 2 && p.epi != EG3D_EPI_ATOMIC && p.patch_rows == 4 && ((p.Ck / 16) % 2) == 0 && p.Ck / 16 >= 4 && p.rgb_out == nullptr;
    if (p.ksplit > 1 && !khalves && (p.epi != EG3D_EPI_ATOMIC || p.ksplit > p.Ck / 16 || p.ksplit > 65535)) return 0;     // every slice owns >= 1 chunk
    if (p.patch_rows != 0 && p.patch_rows != 8 && p.patch_rows != 4 && p.patch_rows != 2) return 0;
    if (p.patch_rows == 4 || p.patch_rows == 2) {         // the half / quarter-height patches are instantiated for the fused 3x3 launches
        if (p.epi == EG3D_EPI_ATOMIC) return 0;
        for (int c = 0; c < p.ncls; ++c) if (p.cls[c].ntaps != 9) return 0;
    }
    if (p.epi == EG3D_EPI_FWD && !eg3d_act_is_pwl(p.act)) return 0;
    if (p.rgb_out != nullptr && (p.epi != EG3D_EPI_FWD || p.Nc != BN || p.ncls != 1 || !p.rgb_w || !p.rgb_s || (p.rgb_ldw & 3) || p.rgb_ldw < p.Nc || (p.rgb_nout != 0 && p.rgb_nout != 3 && p.rgb_nout != 4) || p.patch_rows == 4 || p.patch_rows == 2 || p.cls[0].ntaps != 9)) return 0;
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
    const void* ptrs[] = {p.out, p.addend, p.xin, p.out_scale, p.bias, p.act_bwd.d, p.act_bwd.bias, p.rgb_w, p.rgb_s, p.rgb_bias, p.rgb_out};
    for (const void* q : ptrs)
        if (q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15)) return EG3D_ERR_UNSUPPORTED;
    if (p.epi == EG3D_EPI_FWD && p.noise && !p.noise_strength) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND_V2(det, p); EG3D_DET_COMMIT(det);
    // classes with the same tap count go into one launch (consecutive classes only: 9 | 4,2,2,1 | 1 ...)
    int c = 0;
    while (c < p.ncls) {
        int e = c;
        int max_tiles = 0;
        while (e < p.ncls && p.cls[e].ntaps == p.cls[c].ntaps) {
            const int t = p.N * eg3d_cdiv(p.cls[e].Ha, p.patch_rows == 4 || p.patch_rows == 2 ? p.patch_rows : PH) * eg3d_cdiv(p.cls[e].Wa, PW) * (p.Nc / BN);
            max_tiles = std::max(max_tiles, t);
            ++e;
        }
        int rc;
        switch (p.cls[c].ntaps) {
            case 9:
                if (p.patch_rows == 2) rc = p.products == 1 ? launch_v2<9, false, false, 1>(p, c, e - c, max_tiles, st, 10) : launch_v2<9, true, false, 1>(p, c, e - c, max_tiles, st, 9);
                else if (p.patch_rows == 4 && p.ksplit == 2 && p.epi != EG3D_EPI_ATOMIC)
                    rc = p.products == 1 ? launch_v2<9, false, false, 2, false, 2>(p, c, e - c, max_tiles, st, 14) : launch_v2<9, true, false, 2, false, 2>(p, c, e - c, max_tiles, st, 13);
                else if (p.patch_rows == 4) rc = p.products == 1 ? launch_v2<9, false, false, 2>(p, c, e - c, max_tiles, st, 8) : launch_v2<9, true, false, 2>(p, c, e - c, max_tiles, st, 7);
                else if (p.epi == EG3D_EPI_ATOMIC) rc = p.products == 1 ? launch_v2<9, false, true>(p, c, e - c, max_tiles, st, 6) : launch_v2<9, true, true>(p, c, e - c, max_tiles, st, 5);
                else if (p.rgb_out != nullptr) rc = p.products == 1 ? launch_v2<9, false, false, 4, true>(p, c, e - c, max_tiles, st, 11) : launch_v2<9, true, false, 4, true>(p, c, e - c, max_tiles, st, 12);
                else rc = p.products == 1 ? launch_v2<9, false>(p, c, e - c, max_tiles, st, 4) : launch_v2<9>(p, c, e - c, max_tiles, st, 0);
                break;
            case 4: rc = launch_v2<4>(p, c, e - c, max_tiles, st, 1); break;
            case 2: rc = launch_v2<2>(p, c, e - c, max_tiles, st, 2); break;
            default: rc = launch_v2<1>(p, c, e - c, max_tiles, st, 3); break;
        }
        if (rc != EG3D_OK) return rc;
        c = e;
    }
    EG3D_DET_END(det);
    return EG3D_OK;
}

extern "C" int64_t eg3d_split_activation_bytes(int N, int H, int W, int C) { return (int64_t)N * H * W * C * 4; }

extern "C" int eg3d_split_activation(const float* x, const float* in_scale, const float* x_amax, const float* s_amax, void* image, float* scale_out,
                                     int N, int H, int W, int C, int ldx, void* stream) {
    if (!x || !x_amax || !image || !scale_out || N <= 0 || H <= 0 || W <= 0 || C < 8 || (C & 7) || (ldx & 3) || ldx < C) return EG3D_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(image) & 15)) return EG3D_ERR_UNSUPPORTED;
    const int64_t pix = (int64_t)N * H * W;
    if (C % 128 == 0 || C == 64) {                  // through LDS: channel-major loads, pixel-major stores
        const dim3 grid((unsigned)((pix + 63) / 64), C % 128 == 0 ? C / 128 : 1);
        if (C % 128 == 0)
            hipLaunchKernelGGL(split_act_lds_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, x, in_scale, x_amax, s_amax, reinterpret_cast<f16x8*>(image), scale_out, N, H * W, C, ldx);
        else
            hipLaunchKernelGGL(split_act_lds_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, in_scale, x_amax, s_amax, reinterpret_cast<f16x8*>(image), scale_out, N, H * W, C, ldx);
        EG3D_LAUNCH_CHECK();
        return EG3D_OK;
    }
    const int pb = (int)((pix + 255) / 256), noct = C / 8;
    int gy = 1;                                     // octet groups: >= 4 octets (one 128-byte line of the pixel's row) per thread, ~2048 blocks
    while (gy * 2 <= noct / 4 && pb * gy < 2048) gy *= 2;
    hipLaunchKernelGGL(split_act_kernel, dim3((unsigned)pb, gy), dim3(256), 0, (hipStream_t)stream, x, in_scale, x_amax, s_amax,
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

extern "C" int eg3d_split_weights_batched(const eg3d_split_w_batch* b, void* stream) {
    if (!b || b->n <= 0 || b->n > EG3D_SPLIT_W_BATCH_MAX) return EG3D_ERR_INVALID;
    int64_t big = 0;
    for (int i = 0; i < b->n; ++i) {
        const eg3d_split_w_item& it = b->items[i];
        if (!it.w || !it.amax || !it.image || !it.scale_out || it.O <= 0 || it.I < 16 || (it.I & 15) || it.T <= 0 || it.w_row != it.T * it.I) return EG3D_ERR_INVALID;
        if ((reinterpret_cast<uintptr_t>(it.w) & 15) || (((int64_t)it.O * it.w_row) & 3)) return EG3D_ERR_UNSUPPORTED;
        big = std::max<int64_t>(big, (int64_t)it.O * it.T * (it.I / 8));
    }
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(256, (big + 1023) / 1024));
    hipLaunchKernelGGL(absmax_batched_kernel, dim3(blocks, b->n), dim3(256), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(split_w_batched_kernel, dim3(blocks, b->n), dim3(256), 0, (hipStream_t)stream, *b);
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
