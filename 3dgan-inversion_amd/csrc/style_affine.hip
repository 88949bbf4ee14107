// All style affines of a synthesis network in one launch.
//
// Every modulated layer owns a FullyConnectedLayer(w_dim -> in_channels, bias_init=1) that maps the layer's latent row to its
// styles (training/networks_stylegan2.py:98-108 conv layers, :129-137 toRGB with its extra weight_gain).  Per layer that is a
// [N,512]x[512,C] GEMM plus two scalings -- ~5 launches forward+backward for 26 layers, each far below a microsecond of work.
// Here the bank of layers is one memory-bound pass over the ~18 MB of affine weights:
//   fwd:  styles_l[n, j] = ( sum_k ws[n, wrow_l, k] * (W_l[j,k] * wgain_l) + b_l[j] * bgain_l ) * post_l
//   bwd:  dws[n, wrow_l, k] += sum_j dstyles_l[n, j] * post_l * (W_l[j,k] * wgain_l)
//         dW_l[j,k] = sum_n dstyles_l[n,j] * post_l * wgain_l * ws[n,wrow_l,k],  db_l[j] = sum_n dstyles_l[n,j] * post_l * bgain_l
//         (trainable affines: the pivotal-tuning phase)
#include "common.h"
#include "det.h"

namespace {

constexpr int ROWS_PER_BLOCK_FWD = 4;     // one wave per output row
constexpr int ROWS_PER_BLOCK_BWD = 32;

__device__ __forceinline__ int find_layer(const eg3d_style_bank& b, int tile, const int rows_per_block, int& row0) {
    int l = 0, t0 = 0;
    for (; l < b.nlayers; ++l) {
        const int nt = (b.layers[l].C + rows_per_block - 1) / rows_per_block;
        if (tile < t0 + nt) break;
        t0 += nt;
    }
    row0 = (tile - t0) * rows_per_block;
    return l;
}

__global__ void __launch_bounds__(256) style_affine_fwd_kernel(const eg3d_style_bank b) {
    int row0;
    const int l = find_layer(b, blockIdx.x, ROWS_PER_BLOCK_FWD, row0);
    if (l >= b.nlayers) return;
    const eg3d_style_layer& ly = b.layers[l];
    const int lane = threadIdx.x & 63, j = row0 + (threadIdx.x >> 6);
    if (j >= ly.C) return;
    const float* wrow = ly.weight + (int64_t)j * b.D;
    const float bias = ly.bias ? ly.bias[j] * ly.bgain : 0.f;
    for (int n = 0; n < b.N; ++n) {
        const float* x = b.ws + ((int64_t)n * b.L + ly.wrow) * b.D;
        float acc = 0.f;
        for (int k = lane * 4; k < b.D; k += 256) {
            const float4 w = *reinterpret_cast<const float4*>(wrow + k);
            const float4 v = *reinterpret_cast<const float4*>(x + k);
            acc += v.x * (w.x * ly.wgain);
            acc += v.y * (w.y * ly.wgain);
            acc += v.z * (w.z * ly.wgain);
            acc += v.w * (w.w * ly.wgain);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) ly.out[(int64_t)n * ly.C + j] = (acc + bias) * ly.post;
    }
}

// demodulation coefficients of every conv layer of the bank: one wave per (layer, n, o)
__global__ void __launch_bounds__(256) style_demod_fwd_kernel(const eg3d_style_bank b) {
    int blk = blockIdx.x, l = 0;
    for (; l < b.nlayers; ++l) {
        const int nb = b.layers[l].wsq ? (b.N * b.layers[l].Co + 3) / 4 : 0;
        if (blk < nb) break;
        blk -= nb;
    }
    if (l >= b.nlayers) return;
    const eg3d_style_layer& ly = b.layers[l];
    const int wid = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= b.N * ly.Co) return;
    const int n = wid / ly.Co, o = wid - n * ly.Co;
    const float* s = ly.out + (int64_t)n * ly.C;
    const float* wq = ly.wsq + (int64_t)o * ly.C;
    float acc = 0.f;
    for (int k = lane; k < ly.C; k += 64) { const float sv = s[k]; acc += sv * sv * wq[k]; }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) ly.d[wid] = 1.0f / sqrtf(acc + 1e-8f);
}

// dout_extra[n,k] += -s[n,k] * sum_o dd[n,o] d[n,o]^3 wsq[o,k]:  block = 64 k-columns x 4 o-slices, (k-chunk, n, o-split) per layer
__global__ void __launch_bounds__(256) style_demod_bwd_kernel(const eg3d_style_bank b) {
    __shared__ float part[4][64];
    int blk = blockIdx.x, l = 0, osplit = 1, kchunks = 1;
    for (; l < b.nlayers; ++l) {
        const eg3d_style_layer& q = b.layers[l];
        int nb = 0;
        if (q.dd != nullptr && q.wsq != nullptr) {
            osplit = max(1, min(q.Co / 16, 16));
            kchunks = (q.C + 63) / 64;
            nb = kchunks * b.N * osplit;
        }
        if (blk < nb) break;
        blk -= nb;
    }
    if (l >= b.nlayers) return;
    const eg3d_style_layer& ly = b.layers[l];
    const int kc = blk % kchunks, n = (blk / kchunks) % b.N, oz = blk / (kchunks * b.N);
    const int k = kc * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int per = (ly.Co + osplit - 1) / osplit;
    const int o_beg = oz * per, o_end = min(ly.Co, o_beg + per);
    float acc = 0.f;
    if (k < ly.C) {
        for (int o = o_beg + sl; o < o_end; o += 4) {
            const float dv = ly.d[(int64_t)n * ly.Co + o];
            acc = fmaf(ly.dd[(int64_t)n * ly.Co + o] * dv * dv * dv, ly.wsq[(int64_t)o * ly.C + k], acc);
        }
    }
    part[sl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (sl == 0 && k < ly.C) {
        const float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        eg3d_acc(ly.dout_extra + (int64_t)n * ly.C + k, -ly.out[(int64_t)n * ly.C + k] * t);
    }
}

__global__ void __launch_bounds__(128) style_affine_bwd_kernel(const eg3d_style_bank b) {
    int row0;
    const int l = find_layer(b, blockIdx.x, ROWS_PER_BLOCK_BWD, row0);
    if (l >= b.nlayers) return;
    const eg3d_style_layer& ly = b.layers[l];
    if (ly.dout == nullptr && ly.dout_extra == nullptr) return;
    const int rows = min(ROWS_PER_BLOCK_BWD, ly.C - row0);
    __shared__ float coef[ROWS_PER_BLOCK_BWD];
    for (int n = 0; n < b.N; ++n) {
        __syncthreads();
        if (threadIdx.x < rows) {
            const int64_t j = (int64_t)n * ly.C + row0 + threadIdx.x;
            coef[threadIdx.x] = ((ly.dout ? ly.dout[j] : 0.f) + (ly.dout_extra ? ly.dout_extra[j] : 0.f)) * ly.post;
        }
        __syncthreads();
        if (b.dws == nullptr) continue;              // latent frozen (pivotal tuning): only the weight gradients below are wanted
        float* dst = b.dws + ((int64_t)n * b.L + ly.wrow) * b.D;
        for (int k = threadIdx.x * 4; k < b.D; k += 512) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            // eight rows' loads in flight per trip (with the trip count a run-time value the loop stayed rolled: one 16-byte load per ~1 us of latency)
            for (int r0 = 0; r0 < rows; r0 += 8) {
                float4 w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(ly.weight + (int64_t)(row0 + min(r0 + u, rows - 1)) * b.D + k);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float c = r0 + u < rows ? coef[r0 + u] : 0.f;
                    acc.x += c * (w[u].x * ly.wgain);
                    acc.y += c * (w[u].y * ly.wgain);
                    acc.z += c * (w[u].z * ly.wgain);
                    acc.w += c * (w[u].w * ly.wgain);
                }
            }
            eg3d_acc(dst + k + 0, acc.x);
            eg3d_acc(dst + k + 1, acc.y);
            eg3d_acc(dst + k + 2, acc.z);
            eg3d_acc(dst + k + 3, acc.w);
        }
    }
}

// Gradients of trainable affines: the block owns rows [row0, row0+32) of its layer, so plain stores.
__global__ void __launch_bounds__(128) style_affine_wgrad_kernel(const eg3d_style_bank b) {
    int row0;
    const int l = find_layer(b, blockIdx.x, ROWS_PER_BLOCK_BWD, row0);
    if (l >= b.nlayers) return;
    const eg3d_style_layer& ly = b.layers[l];
    if ((ly.dweight == nullptr && ly.dbias == nullptr) || (ly.dout == nullptr && ly.dout_extra == nullptr)) return;
    const int rows = min(ROWS_PER_BLOCK_BWD, ly.C - row0);
    auto coef = [&](int n, int r) {
        const int64_t j = (int64_t)n * ly.C + row0 + r;
        return ((ly.dout ? ly.dout[j] : 0.f) + (ly.dout_extra ? ly.dout_extra[j] : 0.f)) * ly.post;
    };
    if (ly.dbias != nullptr && threadIdx.x < rows) {
        float s = 0.f;
        for (int n = 0; n < b.N; ++n) s += coef(n, threadIdx.x);
        ly.dbias[row0 + threadIdx.x] = s * ly.bgain;
    }
    if (ly.dweight == nullptr) return;
    for (int k = threadIdx.x * 4; k < b.D; k += 512) {
        for (int r = 0; r < rows; ++r) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int n = 0; n < b.N; ++n) {
                const float4 x = *reinterpret_cast<const float4*>(b.ws + ((int64_t)n * b.L + ly.wrow) * b.D + k);
                const float c = coef(n, r) * ly.wgain;
                acc.x += c * x.x; acc.y += c * x.y; acc.z += c * x.z; acc.w += c * x.w;
            }
            *reinterpret_cast<float4*>(ly.dweight + (int64_t)(row0 + r) * b.D + k) = acc;
        }
    }
}

int check_bank(const eg3d_style_bank* pb, bool bwd) {
    if (!pb) return EG3D_ERR_INVALID;
    const eg3d_style_bank& b = *pb;
    if (b.nlayers < 1 || b.nlayers > EG3D_STYLE_BANK_MAX || b.N < 1 || b.L < 1 || b.D < 4 || (b.D & 3)) return EG3D_ERR_INVALID;
    bool wants_wgrad = false;
    for (int l = 0; l < b.nlayers && l < EG3D_STYLE_BANK_MAX; ++l) wants_wgrad |= b.layers[l].dweight != nullptr || b.layers[l].dbias != nullptr;
    if (!b.ws || (bwd && !b.dws && !wants_wgrad)) return EG3D_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(b.ws) & 15) || (bwd && (reinterpret_cast<uintptr_t>(b.dws) & 15))) return EG3D_ERR_UNSUPPORTED;
    for (int l = 0; l < b.nlayers; ++l) {
        const eg3d_style_layer& ly = b.layers[l];
        if (!ly.weight || ly.C < 1 || ly.wrow < 0 || ly.wrow >= b.L) return EG3D_ERR_INVALID;
        if (!bwd && !ly.out) return EG3D_ERR_INVALID;
        if ((reinterpret_cast<uintptr_t>(ly.weight) & 15) || (reinterpret_cast<uintptr_t>(ly.dweight) & 15)) return EG3D_ERR_UNSUPPORTED;
    }
    return EG3D_OK;
}

int total_tiles(const eg3d_style_bank& b, int rows_per_block) {
    int t = 0;
    for (int l = 0; l < b.nlayers; ++l) t += eg3d_cdiv(b.layers[l].C, rows_per_block);
    return t;
}

}  // namespace

extern "C" int eg3d_style_affine_fwd(const eg3d_style_bank* pb, void* stream) {
    if (int rc = check_bank(pb, false)) return rc;
    hipLaunchKernelGGL(style_affine_fwd_kernel, dim3(total_tiles(*pb, ROWS_PER_BLOCK_FWD)), dim3(256), 0, (hipStream_t)stream, *pb);
    int dblocks = 0;
    for (int l = 0; l < pb->nlayers; ++l) {
        const eg3d_style_layer& ly = pb->layers[l];
        if (ly.wsq != nullptr) {
            if (ly.Co < 1 || ly.d == nullptr) return EG3D_ERR_INVALID;
            dblocks += (pb->N * ly.Co + 3) / 4;
        }
    }
    if (dblocks > 0) hipLaunchKernelGGL(style_demod_fwd_kernel, dim3(dblocks), dim3(256), 0, (hipStream_t)stream, *pb);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_style_affine_bwd(const eg3d_style_bank* pb, void* stream) {
    if (int rc = check_bank(pb, true)) return rc;
    int dblocks = 0;
    for (int l = 0; l < pb->nlayers; ++l) {
        const eg3d_style_layer& ly = pb->layers[l];
        if (ly.dd != nullptr && ly.wsq != nullptr) {
            if (ly.Co < 1 || ly.d == nullptr || ly.dout_extra == nullptr || ly.out == nullptr) return EG3D_ERR_INVALID;
            dblocks += ((ly.C + 63) / 64) * pb->N * std::max(1, std::min(ly.Co / 16, 16));
        }
    }
    EG3D_DET_SCOPE(det, stream);
    for (int l = 0; l < pb->nlayers; ++l) {
        const eg3d_style_layer& ly = pb->layers[l];
        if (ly.dd != nullptr && ly.wsq != nullptr) { EG3D_DET_BIND(det, ly.dout_extra, (int64_t)pb->N * ly.C); }
    }
    EG3D_DET_BIND(det, pb->dws, (int64_t)pb->N * pb->L * pb->D);
    EG3D_DET_COMMIT(det);
    if (dblocks > 0) hipLaunchKernelGGL(style_demod_bwd_kernel, dim3(dblocks), dim3(256), 0, (hipStream_t)stream, *pb);
    EG3D_DET_FLUSH(det);          // (the affine backward reads the demodulation part it just accumulated)
    if (pb->dws != nullptr) hipLaunchKernelGGL(style_affine_bwd_kernel, dim3(total_tiles(*pb, ROWS_PER_BLOCK_BWD)), dim3(128), 0, (hipStream_t)stream, *pb);
    EG3D_DET_END(det);
    bool wants_wgrad = false;
    for (int l = 0; l < pb->nlayers; ++l) wants_wgrad |= pb->layers[l].dweight != nullptr || pb->layers[l].dbias != nullptr;
    if (wants_wgrad) hipLaunchKernelGGL(style_affine_wgrad_kernel, dim3(total_tiles(*pb, ROWS_PER_BLOCK_BWD)), dim3(128), 0, (hipStream_t)stream, *pb);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
