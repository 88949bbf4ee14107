// Helpers shared by the ray-level (renderer.hip) and sample-level (decoder_mfma.hip) kernels of the volume renderer.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace eg3d_render {

constexpr int FC = 32;     // features per plane
constexpr int HD = 64;     // decoder hidden width
constexpr int CO = 32;     // decoder colour outputs

// Decoder non-linearities: 64 softplus + 32 sigmoid per sample.  The libm log1pf/expf expand to ~150 instructions each; the
// hardware exp2/log2 forms are ~10 instructions, absolute error < 2e-7 on the result (softplus(x) = max(x,0) + log(1 + exp(-|x|))
// keeps the argument of log in (1,2]).  The ray marcher keeps the libm forms.
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }
// the log argument lies in [1, 2]: the bare v_log_f32 (log2) needs none of __logf's denormal handling
__device__ __forceinline__ float softplus_fast(float x) { return fmaf(0.6931471805599453f, __builtin_amdgcn_logf(1.f + __expf(-fabsf(x))), fmaxf(x, 0.f)); }

// renderer.py:23-53: plane 0 -> (x,y), plane 1 -> (x,z), plane 2 -> (z,x)
__device__ __forceinline__ void plane_uv(int pl, float x, float y, float z, float& u, float& v) {
    if (pl == 0) { u = x; v = y; } else if (pl == 1) { u = x; v = z; } else { u = z; v = x; }
}

// Index convention of the split-lane layout used by the matrix-core decoder: a 32-vector (features / hidden half / colours)
// of ONE sample lives in TWO lanes l and l+32 (h = l>>5), 16 registers each: element(r, h) = (r&3) + 8*(r>>2) + 4*h.
// This is exactly the row map of the 32x32 MFMA result tile, so a layer's output registers are the next layer's B operand
// (k-pairs (e, e+4)) without any cross-lane traffic.
__device__ __forceinline__ constexpr int split_idx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Work-slot -> ray permutation (eg3d_render_params::ray_tile_width).  Slots are handed to XCDs in contiguous runs, so a run of
// 32*H slots covers a 32-pixel-wide column strip of the image instead of full-width rows: the tri-plane texels a run touches
// then span 1/4 of x (128-wide image) on all three planes, which is what lets them stay in the XCD's 4 MB L2.
__device__ __forceinline__ int64_t ray_of_slot(int64_t slot, int R, int W) {
    if (W <= 0 || (W & 31) || R % W) return slot;
    const int64_t n = slot / R;
    const int i = (int)(slot - n * R), H = R / W;
    const int strip = i / (32 * H), within = i - strip * 32 * H;
    return n * R + (int64_t)(within >> 5) * W + strip * 32 + (within & 31);
}

}  // namespace eg3d_render

// sample-level kernels (decoder_mfma.hip), called by the entry points in renderer.hip
int eg3d_decode_rows_fwd(const eg3d_render_params& p, const float* pos, int pos_stride, int64_t M, int64_t rows_per_image, float* sigma, float* rgb,
                         void* stream, int seg_len = 0, int seg_stride = 0, int seg_off = 0);
int eg3d_decode_rows_bwd(const eg3d_render_bwd_params& bp, const float* pos, int64_t row0, int64_t M, int64_t rows_per_image, int64_t samples_per_ray_row,
                         void* stream);
