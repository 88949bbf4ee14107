// out[i][j] += sum_s a[s][i] * b[s][j],  colsum[i] += sum_s a[s][i]   for tall-skinny row matrices (S ~ 1.6 M rows, <= 64 columns).
//
// The decoder-weight gradients of pivotal tuning (training/triplane.py:116-136 under base_coach.py:96-99) are exactly this shape:
//   dW0 = dpre^T feat [64 x 32], db0 = colsum(dpre), dW1 = dout^T hid [33 x 64], db1 = colsum(dout)   over all 1 572 864 samples.
// Library GEMMs are tuned for the opposite aspect ratio (measured on MI355X: 2.2 ms per product and 1.2 ms per column sum); here every
// wave streams row pairs straight into v_mfma_f32_32x32x2_f32 (K = 2 rows per instruction, exact fp32), keeps the whole <= 64 x 64
// result in its accumulators, and the grid reduces once at the end (LDS per block, then one atomic per element and block).
#include "common.h"
#include "det.h"

// a block's partial products meet in LDS (normal build) or go straight to the exact accumulators of the targets (deterministic build)
#if EG3D_DET
#define GRAM_ACC(r_, c_, v_) do { if ((r_) < Ka && (c_) < Kb) eg3d_acc(out + (r_) * Kb + (c_), (v_) * out_scale); } while (0)
#define GRAM_CS(c_, v_) do { if (colsum != nullptr && (c_) < Ka) eg3d_acc(colsum + (c_), (v_) * cs_scale); } while (0)
#else
#define GRAM_ACC(r_, c_, v_) atomicAdd(&red[(r_) * 64 + (c_)], (v_))
#define GRAM_CS(c_, v_) atomicAdd(&red[64 * 64 + (c_)], (v_))
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GRAM_BLOCKS = 1024;
constexpr int UNROLL = 8;            // row pairs in flight per wave and trip

// A_HI / B_HI: columns 32..63 of a / b go through the matrix cores as well; A_EX (only without A_HI): a has 32 + A_EX columns (A_EX <= 8),
// the extra ones are multiplied on the VALU (a row's extra value is one broadcast load).  The two products of the decoder gradients are
// (64 x 32) and (33 x 64): two MFMAs per row pair each instead of four.
template <bool A_HI, bool B_HI, int A_EX>
__global__ void __launch_bounds__(256) rows_gram_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t S, int Ka, int Kb,
                                                        float* __restrict__ out, float* __restrict__ colsum, float out_scale, float cs_scale) {
    __shared__ float red[64 * 64 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31, h = lane >> 5;
    const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    const bool a0 = l < Ka, a1 = A_HI && 32 + l < Ka, b0 = l < Kb, b1 = B_HI && 32 + l < Kb;
    constexpr int NEX = A_EX > 0 ? A_EX : 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f, ex[NEX][2], exs[NEX];
#pragma unroll
    for (int c = 0; c < NEX; ++c) { ex[c][0] = ex[c][1] = 0.f; exs[c] = 0.f; }
    for (int64_t r0 = 2 * UNROLL * w; r0 < S; r0 += 2 * UNROLL * nw) {
        float av[UNROLL][2], bv[UNROLL][2], xv[UNROLL][NEX];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t row = r0 + 2 * u + h;
            const bool ok = row < S;
            const float* ar = a + row * Ka;
            const float* br = b + row * Kb;
            av[u][0] = (ok && a0) ? ar[l] : 0.f;
            av[u][1] = (ok && a1) ? ar[32 + l] : 0.f;
            bv[u][0] = (ok && b0) ? br[l] : 0.f;
            bv[u][1] = (ok && b1) ? br[32 + l] : 0.f;
#pragma unroll
            for (int c = 0; c < NEX; ++c) xv[u][c] = (A_EX > 0 && ok && 32 + c < Ka) ? ar[32 + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], bv[u][0], acc[0][0], 0, 0, 0);
            if (B_HI) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][0], bv[u][1], acc[0][1], 0, 0, 0);
            if (A_HI) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], bv[u][0], acc[1][0], 0, 0, 0);
            if (A_HI && B_HI) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][1], bv[u][1], acc[1][1], 0, 0, 0);
            cs0 += av[u][0];
            cs1 += av[u][1];
            if (A_EX > 0) {
#pragma unroll
                for (int c = 0; c < NEX; ++c) {
                    ex[c][0] = fmaf(xv[u][c], bv[u][0], ex[c][0]);
                    ex[c][1] = fmaf(xv[u][c], bv[u][1], ex[c][1]);
                    exs[c] += xv[u][c];
                }
            }
        }
    }
    // block reduction in LDS, then one global atomic per element
    for (int i = threadIdx.x; i < 64 * 64 + 64; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (A_HI ? 2 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (B_HI ? 2 : 1); ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = j * 32 + l;
                GRAM_ACC(row, col, acc[i][j][r]);
            }
    cs0 += __shfl_xor(cs0, 32);
    cs1 += __shfl_xor(cs1, 32);
    if (h == 0) { GRAM_CS(l, cs0); if (A_HI) GRAM_CS(32 + l, cs1); }
    if (A_EX > 0) {
#pragma unroll
        for (int c = 0; c < NEX; ++c) {
            GRAM_ACC(32 + c, l, ex[c][0]);                                // both half-waves add their rows' share
            if (B_HI) GRAM_ACC(32 + c, 32 + l, ex[c][1]);
            if (l == 0) GRAM_CS(32 + c, exs[c]);                          // every lane of a half-wave holds the same sum: one of them
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int row = i >> 6, col = i & 63;
        if (row < Ka && col < Kb) eg3d_acc(out + row * Kb + col, red[i] * out_scale);
    }
    if (colsum != nullptr && threadIdx.x < Ka) eg3d_acc(colsum + threadIdx.x, red[64 * 64 + threadIdx.x] * cs_scale);
}

}  // namespace

extern "C" int eg3d_rows_gram_scaled(const float* a, const float* b, int64_t S, int Ka, int Kb, float* out, float* colsum, float out_scale, float colsum_scale,
                                     void* stream);
extern "C" int eg3d_rows_gram(const float* a, const float* b, int64_t S, int Ka, int Kb, float* out, float* colsum, void* stream) {
    return eg3d_rows_gram_scaled(a, b, S, Ka, Kb, out, colsum, 1.f, 1.f, stream);
}

extern "C" int eg3d_rows_gram_scaled(const float* a, const float* b, int64_t S, int Ka, int Kb, float* out, float* colsum, float out_scale, float colsum_scale,
                                     void* stream) {
    if (!a || !b || !out || S < 0 || Ka < 1 || Kb < 1) return EG3D_ERR_INVALID;
    if (Ka > 64 || Kb > 64) return EG3D_ERR_UNSUPPORTED;
    if (S == 0) return EG3D_OK;
    const int blocks = (int)std::min<int64_t>(GRAM_BLOCKS, eg3d_cdiv(S, 2 * UNROLL * 4));
    const bool bhi = Kb > 32;
    const int aex = (Ka > 32 && Ka <= 40) ? Ka - 32 : 0;
    auto kern = rows_gram_kernel<true, true, 0>;
    if (aex == 1) kern = bhi ? rows_gram_kernel<false, true, 1> : rows_gram_kernel<false, false, 1>;
    else if (aex > 1) kern = bhi ? rows_gram_kernel<false, true, 8> : rows_gram_kernel<false, false, 8>;
    else if (Ka > 32) kern = bhi ? rows_gram_kernel<true, true, 0> : rows_gram_kernel<true, false, 0>;
    else kern = bhi ? rows_gram_kernel<false, true, 0> : rows_gram_kernel<false, false, 0>;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, out, (int64_t)Ka * Kb); EG3D_DET_BIND(det, colsum, Ka); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, S, Ka, Kb, out, colsum, out_scale, colsum_scale);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
