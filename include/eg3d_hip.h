/*
 * eg3d_hip.h -- C-ABI of libeg3d_hip.so: the MI355X (gfx950) native hot path of EG3D inversion
 * (TriPlaneGenerator.synthesis forward + backward).  Drop-in boundary for the reference's three JIT-built
 * pybind11 CUDA plugins and for the ATen/cuDNN kernels under G.synthesis.
 *
 * Conventions:
 *   - extern "C", plain device pointers + sizes; no torch / C++ types cross the boundary.
 *   - every entry takes the hipStream_t to launch on (as void*), allocates nothing (the caller owns all
 *     buffers, including workspaces), keeps no global mutable state (re-entrant, multi-stream safe).
 *     ONE exception, in the deterministic build only (libeg3d_hip_det.so, -DEG3D_DET=1; csrc/det.h): the exact accumulators live in a
 *     workspace the caller lends with eg3d_det_set_workspace(), and the table of a call's accumulation targets is one process-global
 *     device object updated in stream order -- that build is single-stream and not re-entrant ("Deterministic build").
 *   - return value: 0 = ok, <0 = invalid argument / unsupported configuration (EG3D_ERR_*),
 *     >0 = hipError_t of a failed launch.  No exceptions.
 *   - tensors are fp32 unless a `dtype` argument says otherwise (EG3D_F32 / EG3D_F16 / EG3D_F64).
 *   - activations on the fused path are NHWC ("channels_last"): element (n,y,x,c) at ((n*H+y)*W+x)*ld + c.
 *
 * Each group cites the reference interface it replaces (paths relative to the reference repo root).
 */
#ifndef EG3D_HIP_H
#define EG3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EG3D_OK 0
#define EG3D_ERR_INVALID (-1)      /* bad argument (null pointer, negative size, ...) */
#define EG3D_ERR_UNSUPPORTED (-2)  /* valid request this build has no kernel for        */
#define EG3D_ERR_TOO_LARGE (-3)    /* exceeds int32 indexing, as the reference checks   */
#define EG3D_ERR_WORKSPACE (-4)    /* deterministic build: the call's accumulation targets do not fit the lent workspace */

#define EG3D_F32 0
#define EG3D_F16 1
#define EG3D_F64 2

/* activation ids == the reference's cuda_idx (torch_utils/ops/bias_act.py:23-33) */
#define EG3D_ACT_LINEAR 1
#define EG3D_ACT_RELU 2
#define EG3D_ACT_LRELU 3
#define EG3D_ACT_TANH 4
#define EG3D_ACT_SIGMOID 5
#define EG3D_ACT_ELU 6
#define EG3D_ACT_SELU 7
#define EG3D_ACT_SOFTPLUS 8
#define EG3D_ACT_SWISH 9

int eg3d_abi_version(void);
const char* eg3d_status_string(int status);

/* ------------------------------------------------------------------------------------------------
 * bias_act -- replaces bias_act_plugin.bias_act(x,b,xref,yref,dy,grad,dim,act,alpha,gain,clamp)
 *   torch_utils/ops/bias_act.cpp:36-94 (host), bias_act.cu:27-151 (kernel), bound at bias_act.cpp:100.
 *   grad=0: y = clamp(act(x + b[(i/step_b) % size_b]) * gain)
 *   grad=1: x := incoming dy; y = dx  (uses yref, or xref+b for swish); zero where |yref| >= clamp
 *   grad=2: x := incoming d_dx; `dy` = first-order dy; y = d_x (2nd-order term)
 *   Any dense layout: flat index, bias index from step_b = x.stride(dim)  (bias_act.cpp:77).
 *   Null pointers stand for the reference's empty tensors.  clamp < 0 disables clamping.
 */
int eg3d_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                  int dtype, int64_t numel, int size_b, int step_b, int grad, int act, float alpha, float gain,
                  float clamp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d -- replaces upfirdn2d_plugin.upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,pady0,pady1,flip,gain)
 *   torch_utils/ops/upfirdn2d.cpp:20-102 (host), upfirdn2d.cu:33-204 (kernels), bound at upfirdn2d.cpp:108.
 *   x: [N,C,inH,inW] with arbitrary element strides xs[4] (NCHW or channels_last); f: fp32 [fH,fW] contiguous;
 *   y: [N,C,outH,outW] with strides ys[4];  outW = (inW*upx + padx0 + padx1 - fW + downx) / downx.
 *   Backward w.r.t. x is the same entry with up<->down, !flip and the padding of upfirdn2d.py:258-268.
 */
int eg3d_upfirdn2d(const void* x, const float* f, void* y, int dtype, int N, int C, int inH, int inW,
                   const int64_t xs[4], int fH, int fW, int upx, int upy, int downx, int downy, int padx0,
                   int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                   const int64_t ys[4], void* stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, fp32 in / fp32 out, products on the matrix cores in the arithmetic `precision` selects (exact fp32 MFMA
 * or 16-bit MFMA products of operand pieces: EG3D_PREC_* below) -- replaces the ATen/cuDNN calls made by
 * conv2d_resample / modulated_conv2d (torch_utils/ops/conv2d_resample.py:31-43,114-136;
 * training/networks_stylegan2.py:34-91) and their autograd backward (the dX contract of
 * torch_utils/ops/conv2d_gradfix.py:139-143).  One launch computes, for every output-grid cell (n,ay,ax):
 *     acc[n,ay,ax,o] = sum_t sum_k  in_scale[n,k] * x[n, ay*in_stride+dy[t], ax*in_stride+dx[t], k] * w[o, wtap[t], k]
 * and applies one of the epilogues below at out[n, ay*out_stride+out_py, ax*out_stride+out_px, o].
 * The tap list expresses: 3x3 / 1x1 correlation (forward), the four parity classes of a stride-2 transposed
 * conv (forward of up=2 layers), the stride-2 correlation (their data gradient) and the flipped 3x3 (data gradient
 * of plain layers).  Up to 4 classes (tap lists + output grids) run in one launch.
 */
#define EG3D_EPI_STORE 0   /* out = acc                                                                  */
#define EG3D_EPI_ATOMIC 1  /* out += acc  (split-K; out must be pre-zeroed)                               */
#define EG3D_EPI_FWD 2     /* out = clamp(act(acc*out_scale[n,o] + noise[n,y,x]*strength + bias[o])*gain) (+ addend); act: linear | relu | lrelu */
#define EG3D_EPI_BWD 3     /* ds[n,o] += sum_px acc*xin ; out = acc*out_scale[n,o] (+ addend)             */
#define EG3D_EPI_BWD_ACT 4 /* EPI_BWD followed, in the same epilogue, by the activation backward of the layer that PRODUCED xin
                            * (eg3d_act_bwd below): the value EPI_BWD would store is that layer's dout, xin is its saved output, so
                            *   dy = dout * act'(xin) * gain (0 where |xin| >= clamp);  out = dy * d[n,o];  + the reductions of
                            * eg3d_modconv_epilogue_bwd.  Saves that pass: dout is never written, xin is read once.           */

/* The producing layer's activation backward for EG3D_EPI_BWD_ACT (same contract as eg3d_modconv_epilogue_bwd: linear / lrelu, reduction
 * targets pre-zeroed and accumulated with atomics; every pointer optional). */
typedef struct eg3d_act_bwd {
    const float* d;            /* [N,Nc] demodulation coefficients of that layer (out = dy * d), or null */
    const float* bias;         /* [Nc] or null                                                          */
    const float* noise;        /* [*,Ho,Wo] or null; batch stride noise_nstride (0 = shared)            */
    int64_t noise_nstride;
    const float* noise_strength;
    int32_t act;
    float alpha, gain, clamp;
    float* dbias;              /* [Nc]   += sum dy                                                      */
    float* dd;                 /* [N,Nc] += sum_px dy * (pre_act - bias - noise*strength) / d            */
    float* dnoise;             /* [*,Ho,Wo] += strength * sum_c dy; batch stride dnoise_nstride          */
    int64_t dnoise_nstride;
    float* dstrength;          /* scalar += sum dy * noise                                              */
} eg3d_act_bwd;

typedef struct eg3d_conv_class {
    int32_t Ha, Wa;            /* output grid of this class                              */
    int32_t out_py, out_px;    /* output pixel = (ay*out_stride + out_py, ax*out_stride + out_px) */
    int32_t ntaps;
    int32_t dy[9], dx[9];      /* input pixel  = (ay*in_stride + dy[t], ax*in_stride + dx[t]); OOB reads are zero */
    int32_t wtap[9];           /* weight tap index of tap t                              */
} eg3d_conv_class;

/* Arithmetic of the implicit GEMM.  Operands, accumulators and results are fp32 in every mode.
 *   F32     v_mfma_f32_32x32x2_f32, exact fp32 products.
 *   BF16X6  each operand cut into 3 bf16 terms (x = b0+b1+b2, exact to 2^-24), six bf16 MFMA products accumulated in fp32;
 *           the dropped cross terms are < 2^-23 relative per product, i.e. fp32-equivalent (below fp32 accumulation error).
 *   BF16X3  three products (b0b0+b0b1+b1b0), relative error < 2^-15 per product (between TF32 and fp32).
 *   F16X3   each operand cut into 2 fp16 terms (x = h+l; residual <= max(2^-22 |x|, 2^-25)), three fp16 MFMA products (hh+hl+lh)
 *           accumulated in fp32: fp32-like for operands of magnitude ~2^-4 .. 2^16; operands outside that range must be brought
 *           into it by the caller with a power-of-two `a_scale` (the result is rescaled exactly). */
enum { EG3D_PREC_F32 = 0, EG3D_PREC_BF16X6 = 1, EG3D_PREC_BF16X3 = 2, EG3D_PREC_F16X3 = 3,
       /* F16X1: the high fp16 piece of each (range-normalised) operand only, ONE product, fp32 accumulation and fp32 results -- the
        * arithmetic of the reference's own fp16 layers (SynthesisBlock with use_fp16 and force_fp32=False, networks_stylegan2.py:421-424:
        * the super-resolution head during pivotal tuning), with operands rounded to fp16 (rel. 2^-11) but nothing stored in fp16. */
       EG3D_PREC_F16X1 = 4 };

typedef struct eg3d_conv_params {
    const float* x;            /* [N,Hi,Wi,ldx] NHWC, Ck used channels                   */
    const float* w;            /* w[o*w_row + tap*Ck + k]                                */
    float* out;                /* [N,Ho,Wo,ldo] NHWC, Nc used channels                   */
    int32_t N, Hi, Wi, Ck, ldx;
    int32_t Nc, w_row;
    int32_t Ho, Wo, ldo;
    int32_t in_stride, out_stride;
    int32_t ncls;
    eg3d_conv_class cls[4];
    const float* in_scale;     /* [N,Ck] or null                                         */
    int32_t epi;
    int32_t ksplit;            /* >=1; >1 only with EG3D_EPI_ATOMIC                      */
    const float* out_scale;    /* [N,Nc] or null                                         */
    const float* bias;         /* [Nc] or null                                           */
    const float* noise;        /* [*,Ho,Wo] or null; batch stride noise_nstride (0 = shared) */
    int64_t noise_nstride;
    const float* noise_strength; /* device scalar (may be null when noise is null)       */
    int32_t act;
    float alpha, gain, clamp;
    const float* addend;       /* [N,Ho,Wo,ldo] or null; may alias out                   */
    const float* xin;          /* EPI_BWD: [N,Ho,Wo,ldo] layer input for the style-gradient reduction, or null */
    float* ds;                 /* EPI_BWD: [N,Nc] accumulated with atomics (pre-zeroed), or null */
    int32_t precision;         /* EG3D_PREC_*: how the fp32 products are formed on the matrix cores */
    const float* a_amax;       /* F16X3 only, optional: device scalar max|x| of the A operand; the kernel multiplies A by the power of
                                * two that brings a_amax * a_amax_mul to ~2^14 and divides the result by it (exact).  null = none. */
    float a_amax_mul;
    int32_t ds_replicas;       /* EPI_BWD: ds is [ds_replicas][N,Nc]; workgroup b accumulates into replica b % ds_replicas so that
                                * thousands of tiles do not serialise on the same N*Nc addresses; the caller sums the replicas.
                                * 0 or 1 = a single [N,Nc] buffer. */
    float* out_amax;           /* optional, pre-zeroed device scalar: receives max|out| (atomic max; EPI_STORE / FWD / BWD) -- the operand
                                * range a consumer needs to range-normalise its two-piece fp16 split */
    eg3d_act_bwd act_bwd;      /* EG3D_EPI_BWD_ACT only (xin required; vector epilogue only: eg3d_conv2d_igemm_act_bwd_ok) */
    int32_t w_presplit;        /* F16X3 only: w is the image written by eg3d_split_weight_pieces (same indexing as the fp32 matrix; w_row and
                                * Ck multiples of 4): the loader copies the two fp16 pieces instead of forming them -- the weight-side half of
                                * the split arithmetic, which this kernel is bound by, is then done once per weight instead of once per
                                * workgroup and K-step.  Same bits, same results. */
    int32_t addend_up2;        /* EPI_FWD, vector epilogue, one class, even Ho / Wo: `addend` is the HALF-resolution image [N,Ho/2,Wo/2,ldo] and
                                * what is added is upfirdn2d.upsample2d(addend) with the separable 4-tap filter addend_taps (taps already
                                * normalised and multiplied by the per-axis gain 2): the skip image of the 'skip' architecture
                                * (networks_stylegan2.py:433-436) is up-sampled inside the toRGB launch instead of by its own pass.
                                * EG3D_ERR_UNSUPPORTED when the conditions do not hold. */
    float addend_taps[4];
} eg3d_conv_params;

/* 1 when this launch can run EG3D_EPI_BWD_ACT (aligned rows, channel counts that are multiples of 4, no split-K, tiles within one
 * image); otherwise run EG3D_EPI_BWD and eg3d_modconv_epilogue_bwd separately. */
int eg3d_conv2d_igemm_act_bwd_ok(const eg3d_conv_params* p);

int eg3d_conv2d_igemm_f32(const eg3d_conv_params* p, void* stream);
/* image[4j .. 4j+3] (16 bytes) = the four high fp16 pieces of w[4j .. 4j+3] followed by their four low pieces (h = rtz16(w),
 * l = rne16(w - h)): the operand image eg3d_conv_params::w_presplit expects.  n floats, n % 4 == 0, both pointers 16-byte aligned. */
int eg3d_split_weight_pieces(const float* w, void* image, int64_t n, void* stream);
/* Which tile configuration eg3d_conv2d_igemm_f32 will launch for p: 0 = 128x128x32 (the dominant kernel), 1 = 64x128,
 * 2 = 32x128, 3 = 128x32.  Pure host function (used by bench.py to attribute launch times to kernels). */
int eg3d_conv2d_igemm_config(const eg3d_conv_params* p);

/* ------------------------------------------------------------------------------------------------
 * Pre-split implicit-GEMM convolution (csrc/conv_v2.hip): same contract as eg3d_conv2d_igemm_f32 in F16X3 arithmetic -- one launch
 * computes, for every cell of every class grid,
 *     acc[n,ay,ax,o] = sum_t sum_k  A[n, ay+dy[t], ax+dx[t], k] * W[o, wtap[t], k]            (in_stride 1; OOB reads are zero)
 * and applies epilogue STORE / ATOMIC / FWD / BWD (as above) at out[n, ay*out_stride+out_py, ax*out_stride+out_px, o] -- but both operands are
 * "split images" prepared once by eg3d_split_activation / eg3d_split_weight (two fp16 pieces per value, see conv_v2.hip), so the
 * style modulation, the range normalisation and the fp32 -> 2 x fp16 split are NOT repeated per tile and per tap:
 *   A image [N][2][Ck/8][Hi][Wi][8] fp16 = split( x * in_scale[n,k] * a_scale ),  a_scale = the power of two that brings
 *           max|x| * max|in_scale| to [2^13, 2^14)  (device scalar written by eg3d_split_activation);
 *   W image [wtaps][Ck/16][2][2][Nc][8] fp16 = split( w[o, tap, k] * w_scale )    (eg3d_split_weight; cached per weight version).
 * The result is divided by a_scale * w_scale in the epilogue (exact).  out_amax (optional, pre-zeroed device scalar) receives
 * max|out| (atomic max): the operand range of the next layer's split.
 * Restrictions (eg3d_conv2d_v2_supported): Ck % 16 == 0, Nc % 128 == 0, in_stride 1, classes of 9 / 4 / 2 / 1 taps whose offsets span
 * at most 3 x 3, 16-byte aligned rows; everything else stays on eg3d_conv2d_igemm_f32. */
typedef struct eg3d_conv_v2_params {
    const void* a;  const void* w;
    const float* a_scale;  const float* w_scale;
    float* out;
    int32_t N, Hi, Wi, Ck;
    int32_t Nc, wtaps;
    int32_t Ho, Wo, ldo;
    int32_t in_stride, out_stride;
    int32_t ncls;
    eg3d_conv_class cls[4];
    int32_t epi;
    const float* out_scale;  const float* bias;  const float* noise;
    int64_t noise_nstride;
    const float* noise_strength;
    int32_t act;
    float alpha, gain, clamp;
    const float* addend;  const float* xin;
    float* ds;
    float* out_amax;
    eg3d_act_bwd act_bwd;      /* EG3D_EPI_BWD_ACT only */
    int32_t products;          /* 0 / 3: three products (fp32-equivalent);  1: high pieces only (EG3D_PREC_F16X1)            */
    int32_t ksplit;            /* 0 / 1: none.  > 1 (EG3D_EPI_ATOMIC only, `out` pre-zeroed): the contraction's 16-channel chunks are split over
                                * ksplit workgroups per tile which add their partial tiles with fp32 atomics -- the layers whose grids
                                * cannot fill the chip (128^2 x 256: 128 tiles; 64^2 x 512: 64).
                                * 2 with a fused (non-atomic) epilogue and patch_rows == 4 (Ck / 16 even and >= 4, no rgb_out): the split stays INSIDE
                                * the workgroup -- eight waves, each four-wave half takes half of the chunks and the halves meet in LDS before the
                                * epilogue: for 4-row grids that give every CU at most one workgroup (two waves per SIMD instead of one).  Same
                                * result as ksplit 0 up to the rounding of one extra sum per element; no zero fill, nothing leaves the workgroup */
    int32_t patch_rows;        /* 0 / 8: workgroup tile = 8 x 32 cells x 128 channels.  4: 4 x 32 cells -- twice the workgroups for 3x3 layers whose
                                * 8-row grid leaves CUs idle, fused epilogues intact (nine-tap classes, not with EG3D_EPI_ATOMIC) */
    /* Optional 1x1 head on the finished tile (eg3d_conv2d_v2 with EG3D_EPI_FWD, Nc == 128 = one channel tile per cell, one nine-tap class,
     * patch_rows 0 / 8): the toRGB layer
     * that reads this layer's output next (training/networks_stylegan2.py:338-359, the super-resolution head's last block) evaluated while the
     * output values are still in registers instead of by a launch that reads the tensor again --
     *     rgb_out[n,y,x,o] = clamp(sum_c out[n,y,x,c] * rgb_w[o * rgb_ldw + c] * rgb_s[n * Nc + c] + rgb_bias[o], +-rgb_clamp),  o = 0 .. 3
     * (exact fp32 multiply-adds; rgb_w: four rows, zero rows for padding channels; rgb_bias may be null; rgb_clamp < 0: none).  rgb_out null: off. */
    const float* rgb_w;  const float* rgb_s;  const float* rgb_bias;
    float* rgb_out;            /* [N,Ho,Wo,4] */
    float rgb_clamp;
    int32_t rgb_ldw;
    int32_t rgb_nout;          /* 0 / 4: four outputs; 3: the fourth row of rgb_w is a zero padding row (a hint: the result is the same) */
} eg3d_conv_v2_params;
int eg3d_conv2d_v2_supported(const eg3d_conv_v2_params* p);
int eg3d_conv2d_v2(const eg3d_conv_v2_params* p, void* stream);
/* Wave-split form of that convolution (csrc/conv_v3.hip) for the nine-tap layers whose grids cannot fill the chip with 256-cell x 128-channel
 * tiles (64^2 x 512, 32^2 x 512, and the 128^2 / 256^2 backbone layers at one image per GPU; training/networks_stylegan2.py:34-91,417-461): same
 * operand images, contract and epilogues (STORE / FWD / BWD / BWD_ACT) as eg3d_conv2d_v2, but a workgroup tile is r x 32 cells x 64 channels
 * (r = p->patch_rows: 0 / 4 | 2) and the contraction is split over the w waves of the workgroup (w = p->ksplit: 0 / 4 | 8; 8 only with r = 2)
 * and summed in LDS in wave order -- no atomics, zero fill, slabs or finishing pass; run-to-run deterministic.  The waves share nothing in the
 * main loop (private LDS halo per wave by LDS-DMA, weight fragments straight from global memory into registers): no barrier in it.
 * Restrictions (eg3d_conv2d_v3_supported): Ck % 16 == 0, Nc % 64 == 0, in_stride 1, nine-tap classes spanning at most 3 x 3. */
int eg3d_conv2d_v3_supported(const eg3d_conv_v2_params* p);
int eg3d_conv2d_v3(const eg3d_conv_v2_params* p, void* stream);
/* Weight-streaming split-K form (csrc/conv_ws.hip) for the 3x3 stride-1 layers of the 4^2 .. 16^2 blocks at one image per GPU (16 .. 256 cells x
 * 512 -> 512: 9.4 MB of weights for <= 1.2 GFLOP; forward and data gradient of training/networks_stylegan2.py:34-91): EG3D_EPI_ATOMIC's contract --
 *     out[n, y, x, o] += sum_t sum_k  x[n, y + dy[t], x + dx[t], k] * in_scale[n, k] * W[o, wtap[t], k]        (out pre-zeroed, fp32, NHWC)
 * with one workgroup per (32-channel tile, four 16-channel chunks of the contraction -- one per wave, summed in LDS in wave order --, block of
 * <= 256 cells): every byte of the weight image is fetched by exactly one wave, all of its loads are in flight before its first matrix
 * instruction, and the fp32 activation is modulated, range-normalised (x_amax * x_amax_mul * max|in_scale|) and split into the two fp16 pieces inside the kernel (no operand pass).
 * Arithmetic of eg3d_conv2d_v2 (w: the weight image of eg3d_split_weight, w_scale its scale).  W <= 32, |dy|, |dx| <= 1, Ck % 16 == 0,
 * Nc % 32 == 0, ldx % 4 == 0. */
typedef struct eg3d_conv_ws_params {
    const float* x;            /* [N,H,W,ldx] fp32 */
    const float* in_scale;     /* [N,Ck] or null */
    const float* x_amax;       /* device scalar: max|x| (or a bound) */
    float x_amax_mul;
    const void* w;  const float* w_scale;
    float* out;                /* [N,H,W,ldo] fp32, accumulated with atomics */
    int32_t N, H, W, Ck, ldx;
    int32_t Nc, ldo, wtaps;
    int32_t dy[9], dx[9], wtap[9];
    int32_t products;          /* 0 / 3 | 1 */
    int32_t in_stride;         /* 0 / 1: correlation, |dy|, |dx| <= 1, x is [N,H,W,ldx].  2: the stride-2 adjoint (data gradient of the up layers):
                                * out[n,a,b,o] += sum x[n, 2a + dy[t], 2b + dx[t], k] ..., dy, dx in 0 .. 2, x is [N,Hx,Wx,ldx] (zeros beyond it),
                                * H * W <= 256 output cells */
    int32_t Hx, Wx;            /* in_stride 2 only */
    int32_t out_stride;        /* 0 / 1.  2: the stride-2 TRANSPOSED conv (forward of an up layer, in_stride <= 1): x is [N,H,W,ldx], out is
                                * [N, 2H + 1, 2W + 1, ldo], out[n, 2a + ky, 2b + kx, o] += x[n,a,b,k] in_scale[n,k] W[o, wtap[3 ky + kx], k]  (dy, dx unused);
                                * (H + 1)(W + 1) <= 96 */
} eg3d_conv_ws_params;
int eg3d_conv2d_ws_supported(const eg3d_conv_ws_params* p);
int eg3d_conv2d_ws(const eg3d_conv_ws_params* p, void* stream);
/* Data gradient of that transposed convolution (a stride-2 3x3 correlation; csrc/conv_v2_s2adj.hip) with the contract, epilogues and
 * weight image of eg3d_conv2d_v2, for ONE class of nine taps (dy, dx) = (t / 3, t % 3):
 *     acc[n,a,b,o] = sum_t sum_k  G[n, 2a + dy[t], 2b + dx[t], k] * W[o, wtap[t], k]
 * where the A operand is the PARITY-split image of G written by eg3d_fir44_adjoint_split: [N][2][Ck/8][4][Hi][Wi][8] fp16 with
 * parity image (py, px)[a', b'] = G[2a' + py, 2b' + px] (p->Hi, p->Wi = dimensions of one parity image, >= Ha + 1, Wa + 1). */
int eg3d_conv2d_v2_s2adj_supported(const eg3d_conv_v2_params* p);
int eg3d_conv2d_v2_s2adj(const eg3d_conv_v2_params* p, void* stream);
/* The same data gradient, same operands and epilogues (STORE / BWD / BWD_ACT), in the wave-split decomposition of eg3d_conv2d_v3 for the layers
 * whose grids cannot fill the chip with 256 x 128 tiles (the backbone's up layers at one image per GPU): 128-cell x 64-channel tiles, the
 * contraction -- the list of (parity image, 16-channel chunk) items with 4 / 2 / 2 / 1 taps -- dealt to the four waves of a workgroup in runs of
 * equal cost and summed in LDS in wave order (csrc/conv_v3.hip).  Nc % 64 == 0, Ck % 16 == 0; taps (dy, dx) = (t / 3, t % 3) as above. */
int eg3d_conv2d_v3_s2adj_supported(const eg3d_conv_v2_params* p);
int eg3d_conv2d_v3_s2adj(const eg3d_conv_v2_params* p, void* stream);
/* FIR adjoint of an up-sampling layer fused with the operand split: G = upfirdn2d(dz, outer([1,3,3,1]) / 64, padding 2, gain) of
 * dz [N, 2 Hi, 2 Wi, ldz] NHWC fp32 (C used channels, C % 8 == 0) -- the (2 Hi + 1) x (2 Wi + 1) input of the data gradient above
 * (torch_utils/ops/upfirdn2d.py:258-268 applied to conv2d_resample.py:129) -- written as the four parity images [.][Hi + 1][Wi + 1] in the
 * split layout, range-normalised by the bound max|G| <= gain * max|dz| (dz_amax: device scalar max|dz|).  image:
 * eg3d_fir44_adjoint_split_bytes() bytes; scale_out: device scalar (the image's power-of-two scale). */
int64_t eg3d_fir44_adjoint_split_bytes(int N, int Hi, int Wi, int C);
int eg3d_fir44_adjoint_split(const float* dz, const float* dz_amax, void* image, float* scale_out, int N, int Hi, int Wi, int C, int ldz, float gain,
                             void* stream);
/* Stride-2 3x3 TRANSPOSED convolution on the same split images, all four output parities per workgroup (csrc/conv_v2_up.hip) -- the
 * F.conv_transpose2d of the up-sampling layers (torch_utils/ops/conv2d_resample.py:114-136):
 *     out[n, 2a + py, 2b + px, o] (+)= sum over taps t = 3 ky + kx with ky % 2 == py, kx % 2 == px of
 *                                     A[n, a - ky/2, b - kx/2, k] * W[o, wtap[t], k]                 (OOB reads are zero)
 * for the cells a < Hc, b < Wc (Hc <= Hi + 1, Wc <= Wi + 1) and the output pixels inside Ho x Wo (<= 2 Hi + 1, 2 Wi + 1).  With
 * Hc = Hi, Wc = Wi the launch covers rows / columns 0 .. 2 Hi - 1 / 2 Wi - 1 on perfectly tiled grids; the last output row and column
 * (a = Hi resp. b = Wi: 1-D problems) are then four small tap classes of eg3d_conv2d_igemm_f32.  epi: EG3D_EPI_STORE, or EG3D_EPI_ATOMIC
 * with ksplit workgroups per tile (out pre-zeroed).  Nc % 64 == 0, Ck % 16 == 0; W image with 9 taps. */
typedef struct eg3d_conv_up2_params {
    const void* a;  const void* w;
    const float* a_scale;  const float* w_scale;
    float* out;
    int32_t N, Hi, Wi, Ck, Nc;
    int32_t Hc, Wc;
    int32_t Ho, Wo, ldo;
    int32_t wtap[9];           /* weight tap index of (ky, kx) = (t / 3, t % 3)                                               */
    int32_t epi, products, ksplit;
    int32_t patch_rows;        /* 0 | 8: 8 x 32-cell patches, eight waves, one workgroup per CU; 4: 4 x 32 cells, four waves, tap-row weight ring, two per CU */
} eg3d_conv_up2_params;
int eg3d_conv2d_up2_supported(const eg3d_conv_up2_params* p);
int eg3d_conv2d_up2(const eg3d_conv_up2_params* p, void* stream);
/* Operand preparation.  x: NHWC fp32 [N,H,W,ldx] (C used channels, C % 8 == 0); in_scale [N,C] or null; x_amax / s_amax: device scalars
 * holding max|x| and max|in_scale| (s_amax null: computed from in_scale by the kernel); image: eg3d_split_activation_bytes() bytes; scale_out: device scalar. */
int64_t eg3d_split_activation_bytes(int N, int H, int W, int C);
int eg3d_split_activation(const float* x, const float* in_scale, const float* x_amax, const float* s_amax, void* image, float* scale_out,
                          int N, int H, int W, int C, int ldx, void* stream);
/* w: packed [O][T][I] fp32 with row stride w_row (the forward or adjoint image of eg3d_pack_conv_weight); image: O*T*I*4 bytes. */
int eg3d_split_weight(const float* w, const float* w_amax, void* image, float* scale_out, int O, int I, int T, int w_row, void* stream);
/* Up to EG3D_SPLIT_W_BATCH_MAX dense packed weight matrices (w_row = T * I) in two launches: max|w| into the pre-zeroed `amax` scalars, then
 * the images (pivotal tuning re-splits every weight once per step). */
#define EG3D_SPLIT_W_BATCH_MAX 24
typedef struct eg3d_split_w_item { const float* w; void* image; float* scale_out; float* amax; int32_t O, I, T, w_row; } eg3d_split_w_item;
typedef struct eg3d_split_w_batch { int32_t n; int32_t pad_; eg3d_split_w_item items[EG3D_SPLIT_W_BATCH_MAX]; } eg3d_split_w_batch;
int eg3d_split_weights_batched(const eg3d_split_w_batch* b, void* stream);
/* out (pre-zeroed device scalar) = max(out, max|x|) over n floats. */
int eg3d_absmax(const float* x, int64_t n, float* out, void* stream);

/* Weight-gradient GEMM (PTI phase: grads into generator weights, training/coaches/base_coach.py:96-99):
 *   dw[o, wtap[t], k] += sum_{n,ay,ax} g[n, ay*out_stride+out_py, ax*out_stride+out_px, o]
 *                                       * in_scale[n,k] * x[n, ay*in_stride+dy[t], ax*in_stride+dx[t], k]
 * Same class/tap description as the forward; dw must be pre-zeroed (split over pixels with atomics). */
typedef struct eg3d_wgrad_params {
    const float* x;  const float* g;  float* dw;
    int32_t N, Hi, Wi, Ck, ldx;
    int32_t Nc, w_row;
    int32_t Ho, Wo, ldg;
    int32_t in_stride, out_stride;
    int32_t ncls;
    eg3d_conv_class cls[4];
    const float* in_scale;     /* [N,Ck] or null */
    int32_t psplit;            /* number of pixel slices */
    int32_t precision;         /* EG3D_PREC_F32 (v_mfma_f32_32x32x2_f32), EG3D_PREC_F16X3 (two fp16 pieces per operand, three products) or EG3D_PREC_F16X1 */
    const float* g_amax;       /* F16X3: optional device scalar max|g|; g is scaled by the power of two that brings g_amax * g_amax_mul */
    float g_amax_mul;          /*        to ~2^13 and the result is scaled back (exact).  null = g is used as it is.                   */
} eg3d_wgrad_params;

int eg3d_conv2d_wgrad_f32(const eg3d_wgrad_params* p, void* stream);
/* n launches of eg3d_conv2d_wgrad_f32 (F16X3 / F16X1 only) as ceil(n / EG3D_WGRAD_BATCH_MAX) launches: the weight gradients of a backward
 * pass are independent of each other, and the small layers' launches (a few dozen workgroups, 10 - 25 us) fill the chip only together. */
#define EG3D_WGRAD_BATCH_MAX 5
int eg3d_conv2d_wgrad_batched(const eg3d_wgrad_params* items, int n, void* stream);

/* The same weight gradient for stride-1 layers whose two operands already exist as split images (csrc/conv_wgrad_v2.hip): g = the image of
 * the gradient operand dz [N][2][Co/8][H][W][8] fp16 (what the data gradient of eg3d_conv2d_v2 consumes), x = the image of the layer input
 * times its styles [N][2][Ci/8][H][W][8] (what its forward consumed), both written by eg3d_split_activation or a fused producer:
 *     dw[o, wtap[t], k] += 1 / (g_scale x_scale) * sum_{n,y,x} G[n,y,x,o] * X[n, y + dy[t], x + dx[t], k]        (OOB reads are zero)
 * for ntaps = 9 (3x3, dy / dx in -1 .. 1) or 1.  LDS-DMA staging, transposing LDS reads (ds_read_b64_tr_b16), no per-element VALU work.
 * Co, Ci multiples of 64; dw pre-zeroed (partial tiles are accumulated with atomics).  products: 0 / 3 three products, 1 high pieces only.
 * row_groups: 0 = chosen by the library (about two workgroups per CU). */
typedef struct eg3d_wgrad_v2_params {
    const void* g;  const void* x;
    const float* g_scale;  const float* x_scale;
    float* dw;
    int32_t N, H, W, Co, Ci, w_row;
    int32_t ntaps;
    int32_t dy[9], dx[9], wtap[9];
    int32_t products, row_groups;
    int32_t slabs;             /* 0: partial tiles are ADDED to dw [Co][w_row] with fp32 atomics (dw pre-zeroed).  != 0: no atomics -- workgroup j of a
                                * channel tile STORES its partial tile into slab j of dw [nslab][Co][w_row] (nslab = eg3d_conv2d_wgrad_v2_slabs(p),
                                * every slab fully written, nothing to pre-zero) and the caller sums the slabs in order
                                * (eg3d_weight_grad_finish_slabs): run-to-run deterministic, and cheaper than ~19 M atomics per launch */
} eg3d_wgrad_v2_params;
int eg3d_conv2d_wgrad_v2_supported(const eg3d_wgrad_v2_params* p);
int eg3d_conv2d_wgrad_v2_slabs(const eg3d_wgrad_v2_params* p);      /* number of slabs (= workgroups per channel tile) the launch will write; < 0: error */
int eg3d_conv2d_wgrad_v2(const eg3d_wgrad_v2_params* p, void* stream);
/* The same for an UP-SAMPLING layer (stride-2 3x3 transposed conv, torch_utils/ops/conv2d_resample.py:114-136 under base_coach.py:96-99):
 *     dw[o, wtap[3 ky + kx], k] += 1 / (g_scale x_scale) * sum_{n,a,b} G_p[n, a + (ky >> 1), b + (kx >> 1), o] * X[n, a, b, k],  p = (ky & 1, kx & 1)
 * p->g = the PARITY-split image of the gradient operand as eg3d_fir44_adjoint_split writes it ([N][2][Co/8][4][H + 1][W + 1][8] fp16),
 * p->x = the split image of the layer input times its styles ([N][2][Ci/8][H][W][8]), H x W = the layer's INPUT resolution; ntaps = 9, dy / dx
 * ignored, slabs = 0 (partial tiles added with atomics: dw pre-zeroed). */
int eg3d_conv2d_wgrad_v2_up_supported(const eg3d_wgrad_v2_params* p);
int eg3d_conv2d_wgrad_v2_up(const eg3d_wgrad_v2_params* p, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tall-skinny Gram product for the decoder-weight gradients of pivotal tuning (training/triplane.py:116-136 under
 * base_coach.py:96-99):  out[i][j] += sum_s a[s][i]*b[s][j]  (Ka, Kb <= 64, row-major a [S,Ka], b [S,Kb]),
 * colsum[i] += sum_s a[s][i] (or null).  out / colsum must be pre-zeroed (accumulated with atomics).  Exact fp32 MFMA. */
int eg3d_rows_gram(const float* a, const float* b, int64_t S, int Ka, int Kb, float* out, float* colsum, void* stream);
/* ... with every partial sum multiplied by out_scale / colsum_scale before it is accumulated (the decoder's runtime weight gains: one launch
 * instead of the product + two scaling passes). */
int eg3d_rows_gram_scaled(const float* a, const float* b, int64_t S, int Ka, int Kb, float* out, float* colsum, float out_scale, float colsum_scale,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * filtered_lrelu -- replaces filtered_lrelu_plugin.filtered_lrelu / filtered_lrelu_act_ (torch_utils/ops/filtered_lrelu.cpp:20,217;
 * Python wrapper filtered_lrelu.py:161-274; slow reference :123-155).  Contiguous NCHW, fp32 or fp16 (float accumulation).
 *   t = gain1 * FIR_fu( zero_insert_up(x + b[c]) padded by (px0,px1,py0,py1) )                 size Hm x Wm
 *   a = mode 0: clamp(lrelu(t, slope) * gain)   [mask := da/dt written when mask != null]      mode 1: t * mask
 *   y = gain2 * decimate_down( FIR_fd( a padded/cropped by (qx0,qx1,qy0,qy1) ) )               size Ho x Wo (checked)
 * Filters are dense 2-D fp32 ([fh][fw], null = identity), flip_* as upfirdn2d's flip_filter.  Forward: q = 0, gain1 = up^2,
 * gain2 = 1; gradients: mode 1 with the stages transposed (see inv3d_amd/torch_utils/ops/filtered_lrelu.py). */
typedef struct eg3d_flrelu_params {
    const void* x;  const void* b;  void* y;
    const float* fu;  const float* fd;
    float* mask;               /* [N,C,Hm,Wm] fp32 */
    int32_t dtype, N, C, H, W;
    int32_t fuh, fuw, fdh, fdw;
    int32_t up, down;
    int32_t px0, px1, py0, py1;
    int32_t qx0, qx1, qy0, qy1;
    int32_t Ho, Wo;
    int32_t flip_fu, flip_fd, mode;
    float gain1, gain2;
    float gain, slope, clamp;  /* mode 0 only; clamp < 0 = none */
} eg3d_flrelu_params;
int eg3d_filtered_lrelu(const eg3d_flrelu_params* p, void* stream);
/* The plugin's stand-alone activation `filtered_lrelu_act_` (torch_utils/ops/filtered_lrelu.cpp:217-272; kernel filtered_lrelu.cu:1110-1215),
 * in place on a contiguous [NC, H, W] tensor:  v = x * gain, then
 *   mode 1 (write signs)  v < 0 -> v *= slope (sign 1);  |v| > clamp -> v = +-clamp (sign 2);  the 2-bit signs go to `signs`
 *   mode 2 (read signs)   bit 0 of the sign at (x + sx, y + sy) -> v *= slope;  bit 1 -> v = 0;  outside the sign image: gain only
 *   mode 0                as mode 1 without a sign image.
 * signs: [NC][sH][sW / 4] bytes, element x in bits 2 (x & 3) of byte x >> 2, sW % 4 == 0 (mode 1: sH >= H, sW >= W).  clamp < 0: none. */
int eg3d_filtered_lrelu_act(void* x, uint8_t* signs, int dtype, int NC, int H, int W, int sH, int sW, int sx, int sy, float gain, float slope,
                            float clamp, int mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Style affines of a whole synthesis network in one launch (the per-layer FullyConnectedLayer(w_dim -> in_channels) of
 * training/networks_stylegan2.py:98-108 and :129-137, ~26 tiny GEMMs + scalings per forward in the reference):
 *   fwd:  out_l[n,j] = ( sum_k ws[n,wrow_l,k] * (weight_l[j,k]*wgain_l) + bias_l[j]*bgain_l ) * post_l
 *         d_l[n,o]   = rsqrt( sum_j out_l[n,j]^2 wsq_l[o,j] + 1e-8 )                      (layers with wsq: second launch)
 *   bwd:  dout_extra_l[n,j] += -out_l[n,j] * sum_o dd_l[n,o] d_l[n,o]^3 wsq_l[o,j]        (layers with dd: first launch)
 *         dws[n,wrow_l,k] += sum_j (dout_l[n,j] + dout_extra_l[n,j]) * post_l * (weight_l[j,k]*wgain_l)   (dws pre-zeroed; null = skip)
 *         dweight_l[j,k]   = sum_n (dout_l[n,j] + dout_extra_l[n,j]) * post_l * wgain_l * ws[n,wrow_l,k]  (trainable affines: the
 *         dbias_l[j]       = sum_n (dout_l[n,j] + dout_extra_l[n,j]) * post_l * bgain_l                     pivotal-tuning phase)
 * ws/dws: [N,L,D] fp32, D a multiple of 4; weight_l: [C_l,D] row-major. */
#define EG3D_STYLE_BANK_MAX 32
typedef struct eg3d_style_layer {
    const float* weight;  const float* bias;  float* out;  const float* dout;
    int32_t C, wrow;
    float wgain, bgain, post;
    int32_t Co;                /* demodulation (conv layers; 0 = none): number of output channels of the layer's conv          */
    const float* wsq;          /* [Co, C] sum over taps of w^2 (networks_stylegan2.py:62-65), or null                             */
    float* d;                  /* fwd out: [N, Co] demodulation coefficients rsqrt(sum_k out[n,k]^2 wsq[o,k] + 1e-8)               */
    const float* dd;           /* bwd in : [N, Co] gradient w.r.t. d, or null                                                     */
    float* dout_extra;         /* bwd scratch: [N, C] zeroed; receives the style gradient that flows through d; added to dout    */
    float* dweight;            /* bwd out: [C, D] gradient of `weight` (overwritten), or null                                    */
    float* dbias;              /* bwd out: [C] gradient of `bias` (overwritten), or null                                         */
} eg3d_style_layer;
typedef struct eg3d_style_bank {
    const float* ws;  float* dws;
    int32_t N, L, D, nlayers;
    eg3d_style_layer layers[EG3D_STYLE_BANK_MAX];
} eg3d_style_bank;
int eg3d_style_affine_fwd(const eg3d_style_bank* bank, void* stream);
int eg3d_style_affine_bwd(const eg3d_style_bank* bank, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layer epilogues (NHWC fp32) -- the fused replacement of upfirdn2d + noise add + bias_act after a modulated
 * conv (training/networks_stylegan2.py:87-90,327-329; conv2d_resample.py:129) and of its backward.
 *
 * fwd: out[n,y,x,c] = clamp(act( FIR(z)[n,y,x,c] * d[n,c] + noise[n,y,x]*strength + bias[c] ) * gain)
 *      FIR: fir==null -> identity (z is [N,H,W,C]); else a (fh x fw) filter applied with padding pad0 (top/left) on
 *      z [N,Hz,Wz,C] and multiplied by fir_gain (the on-path case: 4x4, pad 1, gain 4, Hz = H+1).
 *      out_amax (optional, pre-zeroed device scalar) receives max|out| (atomic max): the operand range of the next layer's split image.
 */
int eg3d_modconv_epilogue_fwd(const float* z, float* out, int N, int H, int W, int C, int Hz, int Wz,
                              const float* fir, int fh, int fw, int pad0, float fir_gain, const float* d,
                              const float* noise, int64_t noise_nstride, const float* noise_strength,
                              const float* bias, int act, float alpha, float gain, float clamp, float* out_amax, void* stream);
/* The same epilogue for a SEPARABLE 4-tap FIR given by its host-side taps k4 (2-D filter = outer(k4, k4); the [1,3,3,1] / 8 filter of every
 * up-sampling layer), C % 64 == 0, act in {linear, relu, lrelu}: staged through LDS (csrc/epilogue.hip: upconv_epilogue_kernel).  With
 * split_image (+ split_in_scale [N,C] = the CONSUMER layer's styles, split_scale_out = device scalar; clamp >= 0 required, it is the range
 * bound) the launch also writes that layer's operand image split(out * split_in_scale) exactly as eg3d_split_activation would
 * (eg3d_split_activation_bytes(N, H, W, C) bytes), so the consumer needs no split pass of its own. */
int eg3d_upconv_epilogue_fwd(const float* z, float* out, int N, int H, int W, int C, int Hz, int Wz, const float* k4, int pad0, float fir_gain,
                             const float* d, const float* noise, int64_t noise_nstride, const float* noise_strength, const float* bias, int act,
                             float alpha, float gain, float clamp, float* out_amax, const float* split_in_scale, void* split_image,
                             float* split_scale_out, void* stream);

/* bwd: given dout and the saved layer output `out`:
 *      dy = dout * act'(out) * gain   (0 where |out| >= clamp; derivative keyed on the OUTPUT as bias_act.cu:76,145)
 *      dz[n,y,x,c] = dy * d[n,c]                    (written; d==null -> dy)
 *      dbias[c]   += sum dy                         (if dbias)
 *      dd[n,c]    += sum_px dy * (pre_act - bias[c] - noise*strength)   (if dd; pre_act recovered from out)
 *      dnoise[n?,y,x] += strength * sum_c dy        (if dnoise; batch stride dnoise_nstride, 0 = shared buffer)
 *      dstrength  += sum dy * noise                 (if dstrength)
 *   All reduction targets must be pre-zeroed; they are accumulated with atomics. */
int eg3d_modconv_epilogue_bwd(const float* dout, const float* out, float* dz, int N, int H, int W, int C,
                              const float* d, const float* noise, int64_t noise_nstride,
                              const float* noise_strength, const float* bias, int act, float alpha, float gain,
                              float clamp, float* dbias, float* dd, float* dnoise, int64_t dnoise_nstride,
                              float* dstrength, float* dz_amax, void* stream);
/* dz_amax (optional, pre-zeroed device scalar): receives max|dz| (atomic max) -- the operand range an F16X3 data-gradient conv needs. */

/* Finish of a split-K data gradient (low-resolution layers, where one launch cannot fill 256 CUs without splitting K):
 *   z = conv data-gradient accumulated with EG3D_EPI_ATOMIC;  dx = z * s[n,c] (+ addend);  ds[n,c] += sum_px z * x  (if ds). */
int eg3d_dgrad_finish(const float* z, const float* x, const float* s, const float* addend, float* dx, float* ds, int N, int H,
                      int W, int C, void* stream);
/* The same finish followed, in the same pass, by the activation backward of the layer that produced x (EG3D_EPI_BWD_ACT for the split-K
 * layers): the value eg3d_dgrad_finish would store is that layer's dout;  dz = dout * act'(x) * gain * d, reductions as in
 * eg3d_modconv_epilogue_bwd (targets in `ab`, pre-zeroed), dz_amax optional. */
int eg3d_dgrad_finish_act(const float* z, const float* x, const float* s, const float* addend, float* dz, float* ds, int N, int H,
                          int W, int C, const eg3d_act_bwd* ab, float* dz_amax, void* stream);
/* Data gradient of a 1x1 layer with FOUR (padded) outputs -- toRGB of the super-resolution head, networks_stylegan2.py:338-359 -- fused
 * with the activation backward of the layer that produced x, as one element-wise pass (eg3d_conv2d_igemm_f32 with EG3D_EPI_BWD_ACT does the
 * same through a GEMM with a 4-deep contraction):  z[px,c] = sum_o dy4[px,o] * wa4[c,o];  dout = z * s[n,c] (+ addend);  ds[n,c] += sum_px z * x;
 * then dz / dbias / dd / dnoise / dstrength / max|dz| exactly as eg3d_dgrad_finish_act.  dy4 [N,H,W,4], wa4 [C,4], x / addend / dz [N,H,W,C]. */
int eg3d_torgb_dgrad_act(const float* dy4, const float* wa4, const float* x, const float* s, const float* addend, float* dz, float* ds, int N, int H,
                         int W, int C, const eg3d_act_bwd* act_bwd, float* dz_amax, void* stream);
/* The same pass writing dz as the two-piece fp16 operand image of the data gradient that consumes it (layout and scale of
 * eg3d_split_activation; replaces that pass): the range comes from a bound formed inside the kernel from dy_amax = max|dy4| and
 * addend_amax = max|addend| (device scalars written by the producers of those tensors; addend_amax is required with addend).  dz may be
 * null when nothing else reads the fp32 gradient.  C % 8 == 0. */
int eg3d_torgb_dgrad_act_split(const float* dy4, const float* wa4, const float* x, const float* s, const float* addend, float* dz, float* ds,
                               int N, int H, int W, int C, const eg3d_act_bwd* ab, const float* dy_amax, const float* addend_amax,
                               void* split_image, float* split_scale_out, void* stream);

/* NHWC FIR resampler used on the fused path (skip-image 2x upsample and its adjoint, FIR adjoint of up=2 layers):
 *   same arithmetic as eg3d_upfirdn2d on a channels-last fp32 tensor, float4 over channels (C % 4 == 0),
 *   optional accumulate into y (y += result). */
int eg3d_upfirdn2d_nhwc(const float* x, const float* f, float* y, int N, int C, int inH, int inW, int fH, int fW,
                        int up, int down, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                        int outH, int outW, int accumulate, void* stream);
/* ... and y = result + addend ([N, outH, outW, C] fp32, 16-byte aligned, not y itself; resampling forms only, i.e. not the plain 4x4 FIR): the skip
 * image of a clamped toRGB layer, img = upsample2d(img) + y (training/networks_stylegan2.py:453-455), in one pass instead of an up-sampling
 * pass and an element-wise add. */
int eg3d_upfirdn2d_nhwc_add(const float* x, const float* f, const float* addend, float* y, int N, int C, int inH, int inW, int fH, int fW,
                            int up, int down, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Style / demodulation helpers (training/networks_stylegan2.py:62-67,303,315,354):
 *   wsq[o,k]   = sum_taps w[o,tap,k]^2                               (once per weight update)
 *   demod fwd:  d[n,o] = rsqrt( sum_k s[n,k]^2 * wsq[o,k] + 1e-8 )
 *   demod bwd:  ds[n,k] += sum_o dd[n,o] * ( -d[n,o]^3 * s[n,k] * wsq[o,k] )
 *               dwsq[o,k] += sum_n dd[n,o] * ( -0.5 * d[n,o]^3 * s[n,k]^2 )       (if dwsq)
 */
int eg3d_weight_sqsum(const float* w, float* wsq, int Co, int ntaps, int Ck, void* stream);
/* One pass over a dense conv weight w[O][I][T] (T = kh*kw): wf[o][t*I+i] (forward operand), wa[i][t*O+o] (data-gradient operand),
 * wsq[o][i] = sum_t w^2 (or null).  Used where weights change every step (pivotal tuning). */
int eg3d_pack_conv_weight(const float* w, float* wf, float* wa, float* wsq, int O, int I, int T, void* stream);
/* The same with a per-output-channel scale folded in, w'[o] = w[o] * oscale[o] (an eval-mode BatchNorm folded into the conv that precedes
 * it: scripts/resnet/resnet.py + w_projector.py:62), no wsq; wa may be null (forward image only).  And the gradient of that fold from
 * the packed weight-gradient image g[o][t*Ip + i] (eg3d_conv2d_wgrad_f32 output, Ip >= I padded input channels):
 *   dw[o][i][t] = g[o][t*Ip + i] * oscale[o] (parameter layout),  doscale[o] = sum_{i,t} g[o][t*Ip + i] * w[o][i][t]   (oscale / doscale may be null). */
int eg3d_pack_conv_weight_scaled(const float* w, const float* oscale, float* wf, float* wa, int O, int I, int T, void* stream);
/* eg3d_pack_conv_weight into buffers whose output-channel dimension is padded to O_pad >= O (toRGB: 3 -> 4 channels so that the image
 * is carried with 16-byte pixels): wf has O_pad rows, wa rows of T*O_pad floats (element (i, t, o) at (i*T + t)*O_pad + o).  Only the O real
 * channels are written: the caller zero-fills both buffers once and reuses them while the weights train. */
int eg3d_pack_conv_weight_padded(const float* w, float* wf, float* wa, float* wsq, int O, int I, int T, int O_pad, void* stream);
/* Weight gradient of a demodulated modulated conv, parameter layout, from the packed weight-gradient image g[o][t*I + i] of
 * eg3d_conv2d_wgrad_f32 plus the demodulation path (networks_stylegan2.py:60-63):
 *   dw[o][i][t] = g[o][t*I + i] + 2 w[o][i][t] * sum_n dd[n,o] (-1/2) d[n,o]^3 s[n,i]^2        (dd null: first term only)
 * s [N,I] styles, d [N,O] demodulation coefficients, dd [N,O] their gradient; w, dw [O,I,T]. */
int eg3d_weight_grad_finish(const float* g, const float* w, const float* s, const float* d, const float* dd, float* dw, int N, int O, int I, int T,
                            void* stream);
/* ... with g = the sum, in slab order, of nslab images [O][T*I] that lie slab_stride floats apart (eg3d_wgrad_v2_params::slabs). */
int eg3d_weight_grad_finish_slabs(const float* g, int nslab, int64_t slab_stride, const float* w, const float* s, const float* d, const float* dd, float* dw,
                                  int N, int O, int I, int T, void* stream);
/* eg3d_weight_grad_finish_slabs of up to EG3D_WGF_BATCH_MAX layers in ONE launch (pivotal tuning: the gradients of all conv weights are
 * wanted together, by the optimiser; same arguments per item). */
#define EG3D_WGF_BATCH_MAX 32
typedef struct eg3d_wgf_item {
    const float* g; const float* w; const float* s; const float* d; const float* dd; float* dw;
    int64_t slab_stride;
    int32_t N, O, I, T, nslab;
} eg3d_wgf_item;
int eg3d_weight_grad_finish_batched(const eg3d_wgf_item* items, int n, void* stream);
/* eg3d_pack_conv_weight (O_pad = 0) / eg3d_pack_conv_weight_padded of up to EG3D_PACK_BATCH_MAX layers in ONE launch: during pivotal tuning
 * all generator weights change together once per step.  wa / wsq may be null per item. */
#define EG3D_PACK_BATCH_MAX 40
typedef struct eg3d_pack_item {
    const float* w; float* wf; float* wa; float* wsq;
    int32_t O, I, T, O_pad;
    const float* oscale;       /* [O] per-output-channel scale folded into wf / wa (eg3d_pack_conv_weight_scaled: a folded BatchNorm), or null */
} eg3d_pack_item;
int eg3d_pack_conv_weights_batched(const eg3d_pack_item* items, int n, void* stream);
int eg3d_unpack_weight_grad(const float* g, const float* w, const float* oscale, float* dw, float* doscale, int O, int I, int Ip, int T, void* stream);
int eg3d_demod_fwd(const float* s, const float* wsq, float* d, int N, int Co, int Ck, void* stream);
int eg3d_demod_bwd(const float* s, const float* wsq, const float* d, const float* dd, float* ds, float* dwsq,
                   int N, int Co, int Ck, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Noise-buffer maintenance of the latent projector (training/projectors/w_projector.py:221-237 noise regulariser and its
 * autograd backward; :264-270 renormalisation), all buffers in one launch each.  x[i]: [res[i],res[i]] fp32 device
 * buffers (res a power of two), nbufs <= 32.
 *   regularizer: *reg_out = scale * sum_buffers sum_levels (mean(x*roll(x,1,W))^2 + mean(x*roll(x,1,H))^2) over the avg-pool
 *                pyramid res, res/2, ... (last level <= 8); grad[i] (may be null / contain nulls) = d(*reg_out)/d x[i].
 *   normalize:   x <- (x - mean(x)) * rsqrt(mean((x - mean)^2))   in place.  workspace: 2*nbufs zeroed floats -> two multi-block
 *                launches (moments, apply); null -> one block per buffer in a single launch.
 */
int64_t eg3d_noise_reg_workspace_floats(const int32_t* res, int nbufs);
int eg3d_noise_regularizer(float* const* x, float* const* grad, const int32_t* res, int nbufs, float* workspace,
                           float* reg_out, float scale, void* stream);
int eg3d_noise_normalize(float* const* x, const int32_t* res, int nbufs, float* workspace, void* stream);

/* Low-latency toRGB for small pixel counts (the 4^2 .. 64^2 blocks of the backbone) -- replaces ToRGBLayer.forward
 * (training/networks_stylegan2.py:338-359) + the skip accumulation of SynthesisBlock.forward (:433-436) where eg3d_conv2d_igemm_f32's
 * 32-step contraction loop is all latency:
 *   out[n,p,o] = clamp( sum_c x[n,p,c] s[n,c] w[o,c] + bias[o] ) + addend[n,p,o]
 * with exact fp32 products (v_mfma_f32_32x32x2_f32).  x [N,H*W,ldx] NHWC (C used, C % 8 == 0), w [Cp][w_row] (row per output, Cp % 32 == 0:
 * padded rows are zero), s [N,C], bias [Cp] or null, clamp < 0 = none; addend null | [N,H,W,ldo] | (addend_up2) the half-resolution image
 * [N,H/2,W/2,ldo] added through upsample2d with the separable 4-tap filter addend_taps (as eg3d_conv_params::addend_up2).  out [N,H,W,ldo]. */
typedef struct eg3d_torgb_small_params {
    const float* x;
    const float* w;
    const float* s;
    const float* bias;
    const float* addend;
    float* out;
    int32_t N, H, W, C, Cp;
    int32_t ldx, ldo, w_row;
    int32_t addend_up2;
    float clamp;
    float addend_taps[4];
    /* optional: the producing layer's finishing epilogue inside this launch.  With pre_z set, x is an OUTPUT: pre_z [N,H*W,ldx] holds that layer's
     * split-K sums, x = clamp(pwl(pre_z * pre_d[n,c] + pre_noise[n,p] * *pre_strength + pre_bias[c]) * pre_gain) is formed while the operand loads
     * (eg3d_modconv_epilogue_fwd's arithmetic; pwl = x > 0 ? x : x * pre_slope) and written to x; max|x| goes to x_amax (atomic max, or null). */
    const float* pre_z;
    const float* pre_d;
    const float* pre_bias;
    const float* pre_noise;
    const float* pre_strength;
    float* x_amax;
    int64_t pre_noise_nstride;
    float pre_slope, pre_gain, pre_clamp;
    int32_t pad_;
} eg3d_torgb_small_params;
int eg3d_torgb_small_supported(const eg3d_torgb_small_params* p);
int eg3d_torgb_small_fwd(const eg3d_torgb_small_params* p, void* stream);
/* Large pixel counts (the 128^2 / 256^2 blocks of the backbone: Cp == 96, C % 32 == 0, C <= 256, H*W % 32 == 0, W >= 32, >= 8192 pixels, no
 * pre_z): eg3d_torgb_small_fwd runs the same arithmetic as a persistent streaming kernel (weights resident in LDS, x read once); this
 * reports whether a launch with these parameters takes that form -- callers use it to decide between this entry and the implicit GEMM. */
int eg3d_torgb_mid_supported(const eg3d_torgb_small_params* p);

/* Data gradient of that layer for small pixel counts: dx[n,p,c] = (sum_o dy[n,p,o] wa[c,o]) s[n,c] + addend[n,p,c]; ds[n,c] += the un-scaled
 * sum times xin[n,p,c] (pre-zeroed, optional).  act_on != 0: additionally the activation backward of the layer that produced xin, exactly as
 * EG3D_EPI_BWD_ACT of eg3d_conv2d_igemm_f32 (dx receives that layer's dz; act_bwd's accumulators are filled).  dy [N,H*W,ldg] (Cp used, Cp % 8
 * == 0), wa [C][wa_row] (row per INPUT channel), C % 32 == 0, xin / addend / dx [N,H*W,ldx].  out_amax: optional pre-zeroed max|dx|. */
typedef struct eg3d_torgb_small_bwd_params {
    const float* dy;
    const float* wa;
    const float* s;
    const float* xin;
    const float* addend;
    float* dx;
    float* ds;
    float* out_amax;
    int32_t N, H, W, C, Cp;
    int32_t ldg, ldx, wa_row;
    int32_t act_on;
    int32_t no_mid;              /* != 0: never the streaming form (eg3d_torgb_mid_bwd_supported reports 0): the A/B switch EG3D_TORGB_MID_BWD=0 reaches every launch */
    eg3d_act_bwd act_bwd;
    /* optional (both or neither; needs addend and xin): the addend is the UNFINISHED split-K data gradient z of the layer that consumes x next
     * (eg3d_dgrad_finish not run): this launch adds z * add_scale[n,c] and accumulates add_ds[n,c] += sum_p z[n,p,c] xin[n,p,c] (pre-zeroed). */
    const float* add_scale;
    float* add_ds;
} eg3d_torgb_small_bwd_params;
int eg3d_torgb_small_bwd_supported(const eg3d_torgb_small_bwd_params* p);
int eg3d_torgb_small_bwd(const eg3d_torgb_small_bwd_params* p, void* stream);
/* The streaming form of the data gradient (Cp == 96, H*W % 32 == 0, >= 4096 pixels): waves walk several pixel tiles with the column sums in
 * registers, one set of atomics per workgroup.  Same results up to summation order; reports whether eg3d_torgb_small_bwd takes that form. */
int eg3d_torgb_mid_bwd_supported(const eg3d_torgb_small_bwd_params* p);

/* torch.optim.Adam(betas, eps; no weight decay, no amsgrad) over up to EG3D_ADAM_ITEMS_MAX leaves in one launch (the projector's
 * optimiser, w_projector.py:107-118,256): p, m (exp_avg), v (exp_avg_sq) updated in place from the gradient g + g2 (one of them may be
 * null).  lr and step are DEVICE scalars (a captured graph is replayed for every step index): step holds the number of updates already
 * made as a float and is incremented by the launch when bump_step != 0 (several launches of one optimiser step: set it on the last).
 * normalize != 0: the leaf is then renormalised in place, p <- (p - mean(p)) * rsqrt(var(p)) (w_projector.py:264-270), by a second launch.
 * workspace: 2*n + 1 ZEROED floats (moments of the flagged leaves, retirement counter; left zeroed-counter on exit). */
/* eg3d_early_stop_flag: done = max(done, value <= threshold ? 1 : 0) on the device (sticky; `value` e.g. the LPIPS term of the step). */
#define EG3D_ADAM_ITEMS_MAX 32
typedef struct {
    float* p;
    const float* g;
    const float* g2;
    float* m;
    float* v;
    int64_t n;
    int32_t normalize, pad_;
} eg3d_adam_item;
typedef struct {
    int32_t n, bump_step;
    float beta1, beta2, eps, pad_;
    const float* lr;
    float* step;
    eg3d_adam_item items[EG3D_ADAM_ITEMS_MAX];
    const float* skip;         /* optional device scalar: != 0 -> the launch changes nothing (no update, no step count): the early stop of the
                                * pivotal-tuning loop, which leaves BEFORE the update (training/coaches/single_id_coach.py:68-71), as a device-side
                                * flag, so that a captured step can be replayed while the criterion is checked every step */
} eg3d_adam_list;
int eg3d_adam_step(const eg3d_adam_list* list, float* workspace, void* stream);
int eg3d_early_stop_flag(const float* value, float threshold, float* done, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pose chain of the latent projector (training/projectors/w_projector.py:147-172 with utils/camera_utils.py:201-228 quaternion, :259-273
 * six-dimensional, :241-257,158-188 Euler): pose vector [B,np] (mode 0: quaternion w,x,y,z, np 4; 1: 6-D, np 6; 2: two angles added to pi/2,
 * np 2) + optimisable translation [B,3] + intrinsics [9] -> extrinsic ext [B,16] and conditioning vector cam [B,25] = (ext, intrinsics).
 * jac [B,12,9] (Jacobian of rotation 3x3 + translation 3 with respect to (pose, translation), forward-mode duals) and out12 [B,12] are
 * caller-owned work buffers the backward reads: d_pose [B,np], d_translation [B,3] (either may be null) from d_ext [B,16] and / or d_cam [B,25]. */
int eg3d_pose_chain_fwd(const float* pose, const float* translation, const float* intrinsics, int B, int mode, float radius, float* ext, float* cam,
                        float* jac, float* out12, void* stream);
int eg3d_pose_chain_bwd(const float* jac, const float* d_ext, const float* d_cam, int B, int mode, float* d_pose, float* d_translation, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Volume renderer -- replaces RaySampler.forward (training/volumetric_rendering/ray_sampler.py:24-73) and
 * ImportanceRenderer.forward (renderer.py:143-195: sample_stratified, sample_from_planes/grid_sample, OSGDecoder
 * triplane.py:124-136, MipRayMarcher2 ray_marcher.py:25-57, sample_importance/sample_pdf, unify_samples), fused
 * per ray; nothing but the final per-ray outputs is materialised.
 */
int eg3d_ray_gen_fwd(const float* cam2world, const float* intrinsics, float* origins, float* dirs, int N, int res,
                     void* stream);
/* d_cam2world [N,16], d_intrinsics [N,9] (may be null) are overwritten. */
int eg3d_ray_gen_bwd(const float* cam2world, const float* intrinsics, const float* d_origins, const float* d_dirs,
                     float* d_cam2world, float* d_intrinsics, int N, int res, void* stream);

typedef struct eg3d_render_params {
    const float* planes;       /* [N,Hp,Wp,ldp] NHWC; plane p = channels [p*C, (p+1)*C)    */
    int32_t N, Hp, Wp, ldp, C; /* C = 32 features per plane                                */
    const float* origins;      /* [N,R,3]                                                  */
    const float* dirs;         /* [N,R,3]                                                  */
    int32_t R;                 /* rays per image                                           */
    const float* u1;           /* [N,R,Dc] stratified jitter in [0,1)                      */
    const float* u2;           /* [N*R,Df] importance uniforms                             */
    int32_t Dc, Df;            /* coarse / fine samples per ray (Df may be 0)              */
    float ray_start, ray_end;  /* used when ray_limits == null                             */
    const float* ray_limits;   /* [N,R,2] per-ray (start,end) for the 'auto' mode, or null */
    int32_t disparity;         /* disparity_space_sampling                                 */
    float box_warp;
    int32_t white_back;
    /* decoder (OSGDecoder): runtime gains already folded in by the caller */
    const float* w0;           /* [H,C]   = net.0.weight * lr_mul/sqrt(C)                  */
    const float* b0;           /* [H]     = net.0.bias * lr_mul                            */
    const float* w1;           /* [H,1+Cout] = (net.2.weight * lr_mul/sqrt(H))^T  (transposed) */
    const float* b1;           /* [1+Cout]                                                 */
    int32_t Hdim, Cout;        /* 64, 32                                                   */
    /* outputs */
    float* rgb;                /* [N,R,Cout]                                               */
    float* depth;              /* [N,R]  (unclamped; NaN kept -- finalize clamps)          */
    float* wsum;               /* [N,R]                                                    */
    float* depth_minmax;       /* [2] running global (min,max) of all sample depths; init (+inf,-inf) by the caller -- the pipelined forward
                                * (pos_rows + save_* given) initialises it itself */
    float* fine_depths;        /* [N,R,Df] workspace: importance depths (re-used by backward)*/
    /* training mode (both or neither): the forward keeps (sigma, colour) of every sample for the backward.
     * Row ((n*R + ray)*2 + pass)*D + s, pass 0 = coarse / 1 = fine, D = max(Dc,Df).                                  */
    float* save_sigma;         /* [S]                                                      */
    float* save_rgb;           /* [S,Cout]                                                 */
    int32_t ray_tile_width;    /* locality hint only (any value gives identical results): when the R rays of an image are the
                                * row-major pixels of an image ray_tile_width wide (a multiple of 32), workgroups are assigned
                                * to rays in 32-pixel-wide column strips so that each XCD's L2 sees a compact screen tile and
                                * hence a small tri-plane footprint.  0 = rays in no particular order. */
    float* pos_rows;           /* optional workspace [2, N*R, D, 4] (D = max(Dc,Df)).  With it (and save_sigma / save_rgb / fine_depths) the
                                * forward runs as a pipeline -- sample positions -> tri-plane gather + decoder on the matrix cores
                                * (the sample-level kernel of the backward) -> importance sampling -> decoder -> compositing from the
                                * saved rows -- instead of one fused per-ray kernel with the decoder on the vector ALUs.  null = fused. */
    float* feat_rows;          /* optional [S,32] (row order of save_rgb), pipelined forward only: the interpolated tri-plane feature of every
                                * sample.  With it the gather runs as its own high-occupancy pass (the fused gather + decoder kernels hold
                                * ~140 registers and wait on the scattered texel loads most of the time) and the decoder kernels -- forward
                                * and, given the same buffer in eg3d_render_bwd_params.fwd, backward -- read the rows back coalesced.
                                * Same values either way.  null = gather inside the decoder kernels. */
    int32_t* dbg_inds;         /* optional [N*R, Df, 3] (pipelined and fused forward): per fine sample the bin index torch.searchsorted(cdf, u,
                                * right=True) would return, and the `below` / `above` indices after the clamps (renderer.py:292-295) -- the
                                * integer side of sample_pdf, for index-exact parity tests.  null = not written. */
    int32_t* dbg_ranks;        /* optional [N*R, Dc + Df]: position of coarse sample s (entry s) and fine sample s (entry Dc + s) in the depth-sorted
                                * list of unify_samples (the inverse of the permutation torch.sort(stable) returns, renderer.py:212-222).  */
    float* dbg_cdf;            /* optional [N*R, ns = Dc - 3]: the ray's CDF edges cdf[1 .. ns] (cdf[0] = 0) exactly as the bin search accumulates them  */
} eg3d_render_params;

int eg3d_render_fwd(const eg3d_render_params* p, void* stream);

/* Element counts (floats) of every caller-owned buffer of the renderer entries for a given problem -- the contract a binding in another
 * language sizes its allocations from (the reference's plugins allocate inside ATen; this library never allocates).  Only N, R, Dc, Df and
 * Cout of `p` are read.  S = N*R*2*max(Dc,Df) sample rows.  A count of 0 = that buffer does not exist for this problem. */
typedef struct eg3d_render_sizes {
    int64_t S;                 /* sample rows                                                            */
    int64_t rgb, depth, wsum, depth_minmax, fine_depths;      /* eg3d_render_fwd outputs                  */
    int64_t save_sigma, save_rgb, pos_rows;                   /* training-mode forward (pos_rows optional) */
    int64_t df_rows, df_pos, ag_rows, gc_rows;                /* eg3d_render_bwd                          */
    int64_t dump_dpre, dump_h, dump_dout, dump_feat;          /* decoder-weight gradient operands         */
    int64_t feat_rows;                                        /* optional feature rows (pipelined forward) */
} eg3d_render_sizes;
int eg3d_render_query_sizes(const eg3d_render_params* p, eg3d_render_sizes* out);
/* depth <- clamp(nan_to_num(depth, inf), min, max) with the global min/max (ray_marcher.py:49-50). */
int eg3d_render_finalize(float* depth, const float* depth_minmax, int64_t n, void* stream);

typedef struct eg3d_render_bwd_params {
    eg3d_render_params fwd;    /* same inputs as the forward (fine_depths filled by it)     */
    const float* depth_out;    /* [N,R] finalized depth (to know which rays were clamped)   */
    const float* d_rgb;        /* [N,R,Cout]                                               */
    const float* d_depth;      /* [N,R] or null                                            */
    const float* d_wsum;       /* [N,R] or null                                            */
    /* Plane gradient: the kernel dumps one row per (ray, pass, s) -- row ((n*R + ray)*2 + pass)*D + s, D = max(Dc,Df) --
     * holding dL/d(mean feature)/3 and the sample position; eg3d_triplane_scatter then bins the rows by texel tile and
     * accumulates them in LDS (a direct scatter would be 384 float atomics per sample).  Both null = no plane gradient. */
    float* df_rows;            /* [S,32]                                                   */
    float* df_pos;             /* [S,4] (x,y,z,-); x = NaN marks an absent sample          */
    float* ag_rows;            /* [S,2] workspace: per-sample (colour weight, dL/d sigma) from the ray-level pass */
    float* gc_rows;            /* [S,4] workspace: per-sample (dL/d position, depth); required with d_origins/d_dirs */
    float* d_origins;          /* [N,R,3] overwritten, or null                             */
    float* d_dirs;             /* [N,R,3] overwritten, or null                             */
    /* Decoder-weight gradients (PTI phase): when non-null the kernel dumps, for sample row
     * ((n*R + ray)*2 + pass)*D + s  (pass 0 = coarse, 1 = fine, D = max(Dc,Df); rows of absent samples untouched, so
     * pre-zero the buffers), the four GEMM operands; the caller contracts them with a library GEMM:
     *   d_w0 = dump_dpre^T @ dump_feat,  d_b0 = colsum(dump_dpre),  d_w1t = dump_h^T @ dump_dout,  d_b1 = colsum(dump_dout) */
    float* dump_dpre;          /* [S,64]  */
    float* dump_h;             /* [S,64]  */
    float* dump_dout;          /* [S,33]  */
    float* dump_feat;          /* [S,32]  */
    float* df_amax;            /* optional [1]: max|df_rows| of this call (reset by the call itself).  Hand it to eg3d_triplane_scatter and the
                                * accumulation runs on the 16-bit matrix cores (three products of two-piece operands, the arithmetic of the
                                * convolutions, operands scaled by this maximum); null = exact fp32 products on the fp32 matrix cores */
    /* Decoder-weight gradients contracted INSIDE the sample-level kernel (all four or none; pre-zeroed, accumulated; excludes the dumps):
     *   gram_w0 [64,32] += gram_scale0 * dpre^T feat      gram_b0 [64] += gram_bias_scale * colsum(dpre)
     *   gram_w1 [33,64] += gram_scale1 * dout^T h         gram_b1 [33] += gram_bias_scale * colsum(dout)        (row / entry 0 = sigma)
     * exact fp32 products (v_mfma_f32_32x32x2_f32), as the caller's GEMM over the dumps would do: the 1.0 GB of dump rows is never written. */
    float* gram_w0; float* gram_b0; float* gram_w1; float* gram_b1;
    float gram_scale0, gram_scale1, gram_bias_scale;
} eg3d_render_bwd_params;

int eg3d_render_bwd(const eg3d_render_bwd_params* p, void* stream);

/* eg3d_render_bwd = a ray-level kernel (merge, march, reverse scan -> ag_rows) + a sample-level kernel (gather, decoder
 * forward/backward -> df_rows/df_pos, gc_rows, optional decoder dumps) + a per-ray reduction of gc_rows.
 *
 * d_planes[N,Hp,Wp,ldp] (pre-zeroed) += bilinear-adjoint scatter of the S dumped rows (grid_sample backward w.r.t. the
 * planes, renderer.py:64 under autograd).  rows_per_image = R*2*D.  workspace: eg3d_triplane_scatter_workspace_ints() int32. */
int64_t eg3d_triplane_scatter_workspace_ints(int64_t S, int N, int Hp, int Wp);
/* ray_w, rows_per_ray: optional hint (0, 0 = unknown) -- rays per image row and dumped rows per ray (2 D) of the row layout above; the
 * binning passes then walk the rows in bricks of 16 x 16 rays (fewer bins per block); the result does not depend on it.
 * df_amax: eg3d_render_bwd_params.df_amax of the call that produced df_rows, or null (see there). */
int eg3d_triplane_scatter(const float* df_rows, const float* df_pos, int64_t S, int64_t rows_per_image, float* d_planes, int N,
                          int Hp, int Wp, int ldp, float box_warp, int32_t* workspace, int ray_w, int rows_per_ray, const float* df_amax, void* stream);

/* Decoder-only query (ImportanceRenderer.run_model, renderer.py:197-203; used for density grids):
 *   coords [N,M,3] -> rgb [N,M,Cout], sigma [N,M]. */
int eg3d_sample_decode(const eg3d_render_params* p, const float* coords, int64_t M, float* rgb, float* sigma,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Perceptual-loss network pieces.  The reference evaluates three third-party networks per step:
 * VGG16-LPIPS features (training/projectors/w_projector.py:50-52,112,215-219), torchvision VGG16 features[:15]
 * (training/warping_loss.py:31-37) and lpips.LPIPS(net='alex') (training/coaches/base_coach.py:48,111-112).  Their convolutions
 * go through eg3d_conv2d_igemm_f32 (bias + ReLU in the epilogue); these entry points are the layers in between.  NHWC fp32,
 * C and ldx multiples of 4, 16-byte aligned pointers.
 *
 * Max pooling without padding (torch.nn.MaxPool2d(k, s), floor mode): y[N,Ho,Wo,C] dense, Ho = (H-k)/s+1.  argmax (optional in
 * forward) holds ky*k+kx of the winning tap, one byte per output element; the first maximum in scan order wins. */
int eg3d_maxpool2d_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, int ldx, int k, int s, void* stream);
/* dx[N,H,W,ldx] (every used channel overwritten) from dy[N,Ho,Wo,C] and the forward's argmax; gather form, no atomics. */
int eg3d_maxpool2d_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int H, int W, int C, int ldx, int k, int s, void* stream);
/* LPIPS feature head (lpips.normalize_tensor + the square root of the 1x1 "lin" layer + spatial mean folded into the features):
 *   feat[n*feat_nstride + (pix*C + c)] = scale[c] * x[n,pix,c] / (sqrt(sum_c x^2) + eps) * mul
 * so that sum((feat_a - feat_b)^2) is the layer's LPIPS term when scale = sqrt(lin weight), mul = 1/sqrt(H*W).  scale may be null (1).
 * feat points at this layer's slice of a flat [N, F] feature vector (feat_nstride = F).
 * eps_inside != 0: the normaliser is rsqrt(sum_c x^2 + eps) instead (the stand-in feature pyramid of the C2 workload). */
int eg3d_unit_normalize_fwd(const float* x, const float* scale, float* feat, int N, int HW, int C, int ldx, float mul, float eps,
                            int64_t feat_nstride, int eps_inside, void* stream);
/* dx[N,HW,ldx] = d feat / d x applied to dfeat (same slice addressing as the forward). */
int eg3d_unit_normalize_bwd(const float* x, const float* scale, const float* dfeat, float* dx, int N, int HW, int C, int ldx, float mul,
                            float eps, int64_t feat_nstride, int eps_inside, void* stream);

/* All taps of the feature pyramid in one launch (grid.y = level).  Per level: x [N,HW,ldx], scale (or null), feat = this level's slice of
 * the flat [N, F] vector (written by the forward, read as dfeat by the backward), dx [N,HW,ldx] (backward only). */
#define EG3D_UNIT_LEVELS_MAX 8
typedef struct {
    const float* x;
    const float* scale;
    const float* feat;      /* forward: output slice (written); backward: dfeat slice */
    float* dx;
    int HW, C, ldx;
    float mul;
} eg3d_unit_level;
typedef struct {
    int n, N, eps_inside;
    float eps;
    int64_t feat_nstride;
    eg3d_unit_level levels[EG3D_UNIT_LEVELS_MAX];
} eg3d_unit_levels;
int eg3d_unit_normalize_levels(const eg3d_unit_levels* batch, int bwd, void* stream);

/* Generator image -> feature-net input, one pass (w_projector.py:198-200,215: (img + 1) * 255/2, then F.interpolate(mode='area') to 256^2):
 *   out[n,y,x,c] = mul * mean_{factor x factor block}(img[n,...,c]) + add  for c < 3,  0 for c = 3.
 * img: [N,H,W,4] NHWC (the SR head's 3-channel image is carried with 4-float pixels), out: [N,H/factor,W/factor,4].
 * bwd: dimg[n,Y,X,c] = mul / factor^2 * dout[n,Y/factor,X/factor,c] (c < 3), 0 for c = 3. */
int eg3d_image_prepare_fwd(const float* img, float* out, int N, int H, int W, int factor, float mul, float add, void* stream);
int eg3d_image_prepare_bwd(const float* dout, float* dimg, int N, int H, int W, int factor, float mul, void* stream);
/* Small-channel 3 x 3 convolution (stride 1, zero padding 1) in exact fp32 on the vector ALUs, for layers too small for the matrix-core tiles
 * (the stand-in feature pyramid of inv3d_amd.inversion.StubFeatureNet: 4 -> 16 -> 32 -> 64 channels at 256^2 .. 64^2; the reference's feature
 * networks are torchvision / lpips modules, w_projector.py:50-58):
 *   y = lrelu(conv(x, W)) * gain (act != 0; plain conv otherwise), written at full resolution (y, optional) and / or as its 2 x 2 average (pooled).
 * x: [N,H,W,Ci] NHWC, Ci % 4 == 0, H and W even.  w: packed [Co/G][Ci/4][9 taps (ky*3+kx)][4][G] (correlation taps: the data gradient passes
 * the flipped, channel-transposed weights).  G in {1,2,4} output channels per thread. */
typedef struct eg3d_conv3x3_direct_params {
    const float* x;
    const float* w;
    float* y;
    float* pooled;
    int N, H, W, Ci, Co, G;
    int act;
    float alpha, gain;
    /* input transform (data gradient of a pooled level; act must be 0, pooled null): with ga or gb set, x is the level's saved full-resolution
     * output and the convolved tensor is  0.25 (ga + gb)[n,y/2,x/2,c] * gain * (x > 0 ? 1 : alpha)  -- eg3d_pool2_act_bwd without its launch */
    const float* ga;
    const float* gb;
} eg3d_conv3x3_direct_params;
int eg3d_conv3x3_direct(const eg3d_conv3x3_direct_params* p, void* stream);
/* Backward of `pooled = avg_pool2(lrelu(z) * gain)` towards z, summing the gradients of the pooled tensor's (up to) two consumers:
 *   dz[n,y,x,c] = 0.25 (ga + gb)[n,y/2,x/2,c] * gain * (yref[n,y,x,c] > 0 ? 1 : alpha)   (ga or gb may be null; NHWC, C % 4 == 0). */
int eg3d_pool2_act_bwd(const float* ga, const float* gb, const float* yref, float* dz, int N, int H, int W, int C, float alpha, float gain, void* stream);
/* out[n] (pre-zeroed) += sum_i (a[n,i] - b[n,i])^2 over flat feature vectors [N,F] (F % 4 == 0) -- the projector's per-image distance
 * (w_projector.py:216-219); bwd: da = 2 g[n] (a - b). */
int eg3d_sqdist_fwd(const float* a, const float* b, float* out, int N, int64_t F, void* stream);
int eg3d_sqdist_bwd(const float* a, const float* b, const float* g, float* da, int N, int64_t F, void* stream);

/* Terms of the pivotal-tuning objective (training/coaches/base_coach.py:104-126 calc_loss; depth TV :294-305) as weighted reductions:
 * each forward adds  value * term_scale  to *term and  value * total_scale  to *total (either may be null; both pre-zeroed), each
 * backward multiplies by the incoming scalar gradient g[0] (device) and the term's weight gscale (host).
 *   sqdist_sum: value = sum_i (a[i] - b[i])^2 over n floats (n % 4 == 0, 16-byte aligned);   da = 2 g gscale (a - b)
 *   tv_norm:    value = sum_{y<H-1, x<W-1} (v - v[x+1])^2 + (v - v[y+1])^2 over a [B,H,W] map;  dv = its gradient * g gscale */
int eg3d_sqdist_sum_fwd(const float* a, const float* b, int64_t n, float* term, float term_scale, float* total, float total_scale, void* stream);
int eg3d_sqdist_sum_bwd(const float* a, const float* b, const float* g, float gscale, float* da, int64_t n, void* stream);
int eg3d_tv_norm_fwd(const float* v, int B, int H, int W, float* term, float term_scale, float* total, float total_scale, void* stream);
int eg3d_tv_norm_bwd(const float* v, const float* g, float gscale, float* dv, int B, int H, int W, void* stream);
/* image_raw with 4-float pixels from the rendered feature image (training/triplane.py:84-85, rgb = features[:, :3]):
 * x [P,C] (C % 4 == 0) -> y4 [P,4] = (x0, x1, x2, 0);  bwd: dx [P,C] = (dy4.xyz, 0, ..., 0) (overwritten). */
int eg3d_slice_rgb4_fwd(const float* x, float* y4, int64_t P, int C, void* stream);
int eg3d_slice_rgb4_bwd(const float* dy4, float* dx, int64_t P, int C, void* stream);
/* ... bwd with the gradient of the feature image's OTHER consumer (the SR head's input, triplane.py:87-88) added in the same pass:
 * dx = addend + (dy4.xyz, 0, ..., 0); addend [P,C], may be dx itself. */
int eg3d_slice_rgb4_bwd_add(const float* dy4, const float* addend, float* dx, int64_t P, int C, void* stream);

/* Depth-reprojection geometry of the warping loss (training/warping_loss.py:18-54 + LinePlaneCollision :58-72) per pixel:
 *   xyz = o + d * depth;  hit = intersection of the line (c, xyz - c) with the plane through P0 with normal -c;
 *   q = A hit + b;  uv = (K (q / q_z) - 0.5) * 2.
 * origins / dirs [P,3], depth [P], uv [P,2]; consts [24] = c[3], P0[3], A[9] (row-major), b[3], K[6] (rows 0..1 of the intrinsics).
 * bwd: gradients w.r.t. origins, dirs and depth from d uv (overwritten). */
int eg3d_warp_project_fwd(const float* origins, const float* dirs, const float* depth, const float* consts, float* uv, int64_t P, void* stream);
int eg3d_warp_project_bwd(const float* origins, const float* dirs, const float* depth, const float* consts, const float* duv, float* d_origins,
                          float* d_dirs, float* d_depth, int64_t P, void* stream);

/* F.grid_sample(input, grid, mode='bilinear', padding_mode='zeros', align_corners=False) for a channels-last input -- the feature warp of the
 * depth-reprojection loss (training/warping_loss.py:50; ATen's kernel takes 152 + 218 us there, this one 6 + 8).
 *   input [N,H,W,C] fp32, C % 4 == 0, 16-byte aligned; grid [N,Ho,Wo,2] (x, y) in [-1,1]; out / dout [N,Ho,Wo,C].
 * bwd: dgrid [N,Ho,Wo,2] is overwritten; dinput (optional, [N,H,W,C], pre-zeroed) is accumulated with atomics. */
int eg3d_grid_sample_nhwc_fwd(const float* input, const float* grid, float* out, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int eg3d_grid_sample_nhwc_bwd(const float* input, const float* grid, const float* dout, float* dgrid, float* dinput, int N, int H, int W, int C, int Ho,
                              int Wo, void* stream);

/* Measurement aid (bench.py): a register-only v_mfma_f32_32x32x16_f16 loop on caller-supplied fp16 data -- what the matrix pipe sustains on
 * this chip at its current power / clock state, timed inside the benchmark run.  in: 4096 x 8 fp16 (64 KiB); out: blocks x 256 floats;
 * executes blocks x 4 waves x iters x 24 MFMAs of 32 x 32 x 16.  No reference counterpart. */
int eg3d_probe_mfma_f16(const void* in, float* out, int blocks, int iters, void* stream);

/* ---- deterministic build (csrc/det.h; `make det` -> libeg3d_hip_det.so) ----------------------------------------------------------------
 * Both builds export these four.  In the deterministic build every floating-point atomic of the library is an exact fixed-point
 * accumulation (order-independent: bit-identical results from run to run), the accumulators living in a workspace the caller lends:
 *   eg3d_det_enabled()                 1 in the deterministic build, 0 in the normal one
 *   eg3d_det_workspace_bytes(n)        bytes for calls that accumulate into at most n float targets each (32 bytes per target + 4 KB)
 *   eg3d_det_set_workspace(ws, bytes)  lend it (16-byte aligned; cleared here, synchronises the stream); ws = NULL returns to float atomics.
 *                                      Calls of the library must then come from one stream at a time.
 *   eg3d_det_misses(out)               additions that fell back to a float atomic (target not bound by the call, |v| >= 2^53); synchronises
 * The reference has no such mode (PyTorch's backward kernels accumulate with atomicAdd the same way). */
int eg3d_det_enabled(void);
int64_t eg3d_det_workspace_bytes(int64_t max_elements_per_call);
int eg3d_det_set_workspace(void* workspace, int64_t bytes, void* stream);
int eg3d_det_misses(uint32_t* out, void* stream);
/* *target += sum of values[0..n), every value added by its own thread through the library's accumulation primitive: float atomics in the
 * normal build (order-dependent), the exact accumulator in the deterministic one (the sum is exact up to the final rounding whatever the
 * values' order and range -- tests/test_gpu_det.py checks that against an exact sum). */
int eg3d_det_accumulate(const float* values, int64_t n, float* target, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EG3D_HIP_H */
