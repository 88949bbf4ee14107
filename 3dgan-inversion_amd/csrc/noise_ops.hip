// Noise-buffer maintenance of the latent projector (training/projectors/w_projector.py:221-237, 264-270), fused.
//
// The reference runs, per optimisation step and per noise buffer (13 backbone + 4 SR maps, 4^2 .. 512^2), a pyramid of
// roll / mul / mean / square / avg_pool2d ops, their autograd backward, and a mean/rsqrt renormalisation: ~2500 tiny launches
// per step, which is what bounds the step once the generator is fast.  Here: three multi-block launches compute the regulariser of all
// buffers AND its gradient (pyramid -> moments -> gradient, below), two renormalise all buffers.
//   reg = sum_buffers sum_levels ( mean(x * roll(x,1,W)) ^2 + mean(x * roll(x,1,H)) ^2 ),  levels: res, res/2, ... while res > 8
#include "common.h"
#include "det.h"

namespace {

constexpr int NT = 1024;
constexpr int MAXB = 32;

struct NoiseBufs {
    float* x[MAXB];
    float* g[MAXB];       // gradient out (may be null)
    int res[MAXB];
    int64_t ws_off[MAXB]; // offset (floats) of this buffer's scratch
    int n;
};

__device__ float block_sum(float v, float* red) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < NT / 64) t = red[threadIdx.x];
    if (w == 0) {
        for (int m = 8; m >= 1; m >>= 1) t += __shfl_xor(t, m);
        if (l == 0) red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// ---- regulariser: three multi-block passes ---------------------------------------------------------------------------------------
// Round 2 ran ONE 1024-thread block per buffer over the whole pyramid: the 512^2 map of the SR head made it a 440-465 us kernel on 17 of
// 256 CUs (3.3 MB of data: ~1 us at HBM speed).  Now
//   pass 1 (pyramid):  block = (buffer, band of 64 rows): the 2x2-average pyramid of its band, all levels (bands are aligned to 2^6 rows,
//                      so pooling never crosses a band), written to the scratch;
//   pass 2 (moments):  block = (buffer, level, chunk): sum v * roll_x(v), sum v * roll_y(v) -> one atomic pair per block;
//   pass 3 (gradient): block = (buffer, chunk of level 0): with m_l = the level's two means, the chain rule through the average pools is
//                      g[i] = scale * sum_l 4^-l * 2/n_l * ( mx_l (left + right) + my_l (up + down) ) at the level-l ancestor of i
//                      -- every element independent -- and the buffer's regulariser value from the moments.
constexpr int BAND = 64;                 // rows of level 0 per pyramid block (>= 2^(levels-1) for res <= 512; larger maps: see noise_levels)
constexpr int CHUNK = NT * 4;            // elements per block in passes 2 and 3
constexpr int MAXL = 12;

__host__ __device__ inline int noise_levels(int res) { int nl = 1; for (int r = res; r > 8; r >>= 1) ++nl; return nl; }
// rows of level 0 a pyramid block owns: the whole coarsest level (8 rows) must split evenly over the bands
__host__ __device__ inline int noise_band(int res) { const int b = res / 8; return b < 1 ? res : (b < BAND ? (res < BAND ? res : BAND) : b); }
__host__ __device__ inline int64_t lev_off(int res, int l) {        // offset of level l >= 1 inside a buffer's scratch
    int64_t o = 0;
    for (int k = 1; k < l; ++k) { const int r = res >> k; o += (int64_t)r * r; }
    return o;
}

__global__ void __launch_bounds__(NT) noise_pyramid_kernel(const NoiseBufs B, float* __restrict__ ws, float* __restrict__ sums, float* __restrict__ reg_out) {
    int blk = blockIdx.x, b = 0, band = 0;
    for (; b < B.n; ++b) {
        const int nb = B.res[b] / noise_band(B.res[b]);
        if (blk < nb) { band = blk; break; }
        blk -= nb;
    }
    if (b >= B.n) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) *reg_out = 0.f;
    if (band == 0 && threadIdx.x < 2 * MAXL) sums[b * 2 * MAXL + threadIdx.x] = 0.f;
    const int res = B.res[b], nl = noise_levels(res), R = noise_band(res);
    float* scratch = ws + B.ws_off[b];
    const float* cur = B.x[b];
    int r = res, rows = R, y0 = band * R;            // current level: width r, this band's rows [y0, y0 + rows)
    for (int l = 1; l < nl; ++l) {
        const int h = r >> 1, hrows = rows >> 1, hy0 = y0 >> 1;
        float* nxt = scratch + lev_off(res, l);
        const int sh = __ffs(h) - 1;
        for (int i = threadIdx.x; i < hrows * h; i += NT) {
            const int yy = i >> sh, xx = i - (yy << sh);
            const float2 p0 = *reinterpret_cast<const float2*>(cur + (int64_t)(y0 + 2 * yy) * r + 2 * xx);
            const float2 p1 = *reinterpret_cast<const float2*>(cur + (int64_t)(y0 + 2 * yy + 1) * r + 2 * xx);
            nxt[(int64_t)(hy0 + yy) * h + xx] = ((p0.x + p0.y) + (p1.x + p1.y)) * 0.25f;
        }
        __threadfence_block();
        __syncthreads();
        cur = nxt; r = h; rows = hrows; y0 = hy0;
    }
}

// block -> (buffer, level, chunk) for pass 2; level < 0: not a block of this launch
__device__ __forceinline__ void moments_locate(const NoiseBufs& B, int blk, int& buf, int& lev, int& start) {
    for (buf = 0; buf < B.n; ++buf) {
        const int nl = noise_levels(B.res[buf]);
        for (lev = 0; lev < nl; ++lev) {
            const int r = B.res[buf] >> lev;
            const int nb = (r * r + CHUNK - 1) / CHUNK;
            if (blk < nb) { start = blk * CHUNK; return; }
            blk -= nb;
        }
    }
    lev = -1;
}

__global__ void __launch_bounds__(NT) noise_moments2_kernel(const NoiseBufs B, const float* __restrict__ ws, float* __restrict__ sums) {
    __shared__ float red[32];
    int b, l, start;
    moments_locate(B, blockIdx.x, b, l, start);
    if (l < 0) return;
    const int res = B.res[b], r = res >> l, n = r * r, sh = __ffs(r) - 1;
    const float* v = l == 0 ? B.x[b] : ws + B.ws_off[b] + lev_off(res, l);
    float sx = 0.f, sy = 0.f;
    float c[CHUNK / NT], lx[CHUNK / NT], uy[CHUNK / NT];       // all twelve loads of the block's chunk in flight (a rolled loop waited for each triple)
#pragma unroll
    for (int u = 0; u < CHUNK / NT; ++u) {
        const int i = min(start + threadIdx.x + u * NT, n - 1);
        const int y = i >> sh, xx = i - (y << sh);
        c[u] = v[i];
        lx[u] = v[(y << sh) + ((xx - 1) & (r - 1))];
        uy[u] = v[(((y - 1) & (r - 1)) << sh) + xx];
    }
#pragma unroll
    for (int u = 0; u < CHUNK / NT; ++u)
        if (start + threadIdx.x + u * NT < n) { sx += c[u] * lx[u]; sy += c[u] * uy[u]; }
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    if (threadIdx.x == 0) { eg3d_acc(sums + (b * MAXL + l) * 2, sx); eg3d_acc(sums + (b * MAXL + l) * 2 + 1, sy); }
}

__global__ void __launch_bounds__(NT) noise_grad_kernel(const NoiseBufs B, const float* __restrict__ ws, const float* __restrict__ sums, float* __restrict__ reg_out,
                                                        float scale) {
    int blk = blockIdx.x, b = 0;
    for (; b < B.n; ++b) {
        const int nb = (B.res[b] * B.res[b] + CHUNK - 1) / CHUNK;
        if (blk < nb) break;
        blk -= nb;
    }
    if (b >= B.n) return;
    const int res = B.res[b], nl = noise_levels(res);
    __shared__ float cx[MAXL], cy[MAXL];
    if (threadIdx.x < 16) {                            // lanes 0 .. nl-1 of wave 0 hold one level each
        float part = 0.f;
        if (threadIdx.x < nl) {
            const int r = res >> threadIdx.x;
            const float n = (float)(r * r);
            const float mx = sums[(b * MAXL + threadIdx.x) * 2] / n, my = sums[(b * MAXL + threadIdx.x) * 2 + 1] / n;
            const float w = 2.f / n * scale / (float)(1 << (2 * threadIdx.x));    // 2 m / n_l  x  4^-l (one average pool per level)
            cx[threadIdx.x] = mx * w;
            cy[threadIdx.x] = my * w;
            part = mx * mx + my * my;
        }
        for (int m = 8; m >= 1; m >>= 1) part += __shfl_xor(part, m, 16);
        if (blk == 0 && threadIdx.x == 0) eg3d_acc(reg_out, part * scale);          // this buffer's part of the value
    }
    float* gout = B.g[b];
    if (gout == nullptr) return;
    __syncthreads();
    const float* x0 = B.x[b];
    const float* scratch = ws + B.ws_off[b];
    const int sh0 = __ffs(res) - 1, n0 = res * res;
#pragma unroll 2
    for (int i = blk * CHUNK + threadIdx.x; i < min(n0, (blk + 1) * CHUNK); i += NT) {
        const int y = i >> sh0, xx = i - (y << sh0);
        float g = 0.f;
        const float* v = x0;
        int64_t off = 0;
        for (int l = 0; l < nl; ++l) {
            const int r = res >> l, sh = sh0 - l, yl = y >> l, xl = xx >> l;
            g += cx[l] * (v[(yl << sh) + ((xl - 1) & (r - 1))] + v[(yl << sh) + ((xl + 1) & (r - 1))]) +
                 cy[l] * (v[(((yl - 1) & (r - 1)) << sh) + xl] + v[(((yl + 1) & (r - 1)) << sh) + xl]);
            v = scratch + off;                         // level l + 1
            off += (int64_t)(r >> 1) * (r >> 1);
        }
        gout[i] = g;
    }
}

__global__ void __launch_bounds__(NT) noise_normalize_kernel(const NoiseBufs B) {
    __shared__ float red[32];
    float* x = B.x[blockIdx.x];
    const int n = B.res[blockIdx.x] * B.res[blockIdx.x];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) s += x[i];
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) { float v = x[i] - mean; q += v * v; }
    const float inv = 1.0f / sqrtf(block_sum(q, red) / (float)n);
    for (int i = threadIdx.x; i < n; i += NT) x[i] = (x[i] - mean) * inv;
}

// Multi-block form of the renormalisation (the 256^2 / 512^2 buffers of the SR head make the one-block-per-buffer kernel above a
// 110 us serial tail of every step): pass 1 accumulates sum(x) and sum(x^2) per buffer with one atomic pair per block, pass 2 applies
// (x - mean) * rsqrt(E[x^2] - mean^2)  ==  the reference's  buf -= mean; buf *= rsqrt(mean(buf^2))  (w_projector.py:264-270).
constexpr int NORM_CHUNK = NT * 8;            // elements per block

__device__ __forceinline__ bool norm_locate(const NoiseBufs& B, int blk, int& buf, int& start, int& n) {
    int b0 = 0;
    for (buf = 0; buf < B.n; ++buf) {
        n = B.res[buf] * B.res[buf];
        const int nb = (n + NORM_CHUNK - 1) / NORM_CHUNK;
        if (blk < b0 + nb) { start = (blk - b0) * NORM_CHUNK; return true; }
        b0 += nb;
    }
    return false;
}

__global__ void __launch_bounds__(NT) noise_moments_kernel(const NoiseBufs B, float* __restrict__ ws) {
    __shared__ float red[32];
    int buf, start, n;
    if (!norm_locate(B, blockIdx.x, buf, start, n)) return;
    const float* x = B.x[buf];
    float s = 0.f, q = 0.f;
    for (int i = start + threadIdx.x; i < min(n, start + NORM_CHUNK); i += NT) { const float v = x[i]; s += v; q += v * v; }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) { eg3d_acc(ws + 2 * buf, s); eg3d_acc(ws + 2 * buf + 1, q); }
}

__global__ void __launch_bounds__(NT) noise_apply_norm_kernel(const NoiseBufs B, const float* __restrict__ ws) {
    int buf, start, n;
    if (!norm_locate(B, blockIdx.x, buf, start, n)) return;
    float* x = B.x[buf];
    const float mean = ws[2 * buf] / (float)n;
    const float inv = 1.0f / sqrtf(ws[2 * buf + 1] / (float)n - mean * mean);
    for (int i = start + threadIdx.x; i < min(n, start + NORM_CHUNK); i += NT) x[i] = (x[i] - mean) * inv;
}

// ---- Adam over the projector's leaves (w_projector.py:107-118, 256-270), one launch --------------------------------------------------
// torch's multi-tensor Adam walks 64 K-element chunks with 512-thread blocks: ~20 blocks for the 3.3 MB of noise maps, 45 us, after a
// 19 us multi-tensor add of the regulariser's gradient, and before a moments pass over the same maps.  Here: block = (leaf, 4096
// elements); the gradient is g + g2 (either may be absent), the update is torch.optim.Adam's (no weight decay, no amsgrad), and leaves
// flagged `normalize` leave sum / sum of squares of their NEW values behind for the renormalisation that follows.  The step count lives
// on the device (graph replay): every block reads it on entry, the last block to retire writes it back incremented.
constexpr int ADAM_CHUNK = NT * 4;

__device__ __forceinline__ bool adam_locate(const eg3d_adam_list& A, int blk, int& item, int64_t& start) {
    int b0 = 0;
    for (item = 0; item < A.n; ++item) {
        const int nb = (int)((A.items[item].n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        if (blk < b0 + nb) { start = (int64_t)(blk - b0) * ADAM_CHUNK; return true; }
        b0 += nb;
    }
    return false;
}

__global__ void early_stop_flag_kernel(const float* __restrict__ value, float thr, float* __restrict__ done) {
    if (*value <= thr) *done = 1.0f;
}

__global__ void __launch_bounds__(NT) adam_step_kernel(const eg3d_adam_list A, float* __restrict__ ws) {
    __shared__ float red[32];
    // the three device scalars are read together (one memory round trip in front of the update instead of three in sequence)
    const float skipv = A.skip != nullptr ? *A.skip : 0.f;
    const float t = *A.step + 1.0f;
    const float lr = *A.lr;
    if (skipv != 0.f) return;                                  // (uniform over the grid: nothing is touched, the step count stays)
    int item;
    int64_t start;
    if (adam_locate(A, blockIdx.x, item, start)) {
        const eg3d_adam_item& it = A.items[item];
        const float bc1 = 1.0f - powf(A.beta1, t), bc2s = sqrtf(1.0f - powf(A.beta2, t));
        const float step_size = lr / bc1;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < ADAM_CHUNK / NT; ++u) {
            const int64_t i = start + threadIdx.x + u * NT;
            if (i < it.n) {
                float g = it.g ? it.g[i] : 0.f;
                if (it.g2) g += it.g2[i];
                const float m = it.m[i] + (g - it.m[i]) * (1.0f - A.beta1);
                const float v = it.v[i] * A.beta2 + (1.0f - A.beta2) * g * g;
                const float x = it.p[i] - step_size * m / (sqrtf(v) / bc2s + A.eps);
                it.m[i] = m; it.v[i] = v; it.p[i] = x;
                s += x; q += x * x;
            }
        }
        if (it.normalize) {                  // (block-uniform)
            s = block_sum(s, red);
            q = block_sum(q, red);
            if (threadIdx.x == 0) { eg3d_acc(ws + 2 * item, s); eg3d_acc(ws + 2 * item + 1, q); }
        }
    }
    if (A.bump_step && threadIdx.x == 0) {
        __threadfence();
        unsigned* tick = reinterpret_cast<unsigned*>(ws + 2 * A.n);
        if (atomicAdd(tick, 1u) == gridDim.x - 1) { *A.step = t; *tick = 0u; }
    }
}

__global__ void __launch_bounds__(NT) adam_apply_norm_kernel(const eg3d_adam_list A, const float* __restrict__ ws) {
    int item;
    int64_t start;
    if (A.skip != nullptr && *A.skip != 0.f) return;
    if (!adam_locate(A, blockIdx.x, item, start)) return;
    const eg3d_adam_item& it = A.items[item];
    if (!it.normalize) return;
    const float mean = ws[2 * item] / (float)it.n;
    const float inv = 1.0f / sqrtf(ws[2 * item + 1] / (float)it.n - mean * mean);
#pragma unroll
    for (int u = 0; u < ADAM_CHUNK / NT; ++u) {
        const int64_t i = start + threadIdx.x + u * NT;
        if (i < it.n) it.p[i] = (it.p[i] - mean) * inv;
    }
}

int fill(NoiseBufs& B, float* const* x, float* const* g, const int32_t* res, int nbufs) {
    if (!x || !res || nbufs < 1 || nbufs > MAXB) return EG3D_ERR_INVALID;
    B.n = nbufs;
    int64_t off = 0;
    for (int i = 0; i < nbufs; ++i) {
        if (!x[i] || res[i] < 1 || (res[i] & (res[i] - 1)) || res[i] > 4096) return EG3D_ERR_INVALID;
        B.x[i] = x[i];
        B.g[i] = g ? g[i] : nullptr;
        B.res[i] = res[i];
        B.ws_off[i] = off;
        off += (int64_t)res[i] * res[i];          // >= 2 * (1/4 + 1/16 + ...) of the buffer
    }
    return EG3D_OK;
}

}  // namespace

extern "C" int64_t eg3d_noise_reg_workspace_floats(const int32_t* res, int nbufs) {
    int64_t t = 0;
    for (int i = 0; i < nbufs; ++i) t += (int64_t)res[i] * res[i];      // per buffer: its pyramid (< res^2 / 3), 16-byte aligned slots
    return t + 2 * MAXL * MAXB;                                        // + the per-(buffer, level) moment pairs
}

extern "C" int eg3d_noise_regularizer(float* const* x, float* const* grad, const int32_t* res, int nbufs, float* workspace, float* reg_out, float scale,
                                      void* stream) {
    NoiseBufs B;
    int rc = fill(B, x, grad, res, nbufs);
    if (rc) return rc;
    if (!workspace || !reg_out) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    int64_t pyr = 0;
    int nb1 = 0, nb2 = 0, nb3 = 0;
    for (int i = 0; i < nbufs; ++i) {
        pyr += (int64_t)res[i] * res[i];
        nb1 += res[i] / noise_band(res[i]);
        for (int l = 0; l < noise_levels(res[i]); ++l) nb2 += eg3d_cdiv((int64_t)(res[i] >> l) * (res[i] >> l), CHUNK);
        nb3 += eg3d_cdiv((int64_t)res[i] * res[i], CHUNK);
    }
    float* sums = workspace + pyr;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, sums, 2 * MAXL * MAXB); EG3D_DET_BIND(det, reg_out, 1); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(noise_pyramid_kernel, dim3(nb1), dim3(NT), 0, st, B, workspace, sums, reg_out);     // also zeroes the moments and reg_out
    hipLaunchKernelGGL(noise_moments2_kernel, dim3(nb2), dim3(NT), 0, st, B, workspace, sums);
    EG3D_DET_FLUSH(det);
    hipLaunchKernelGGL(noise_grad_kernel, dim3(nb3), dim3(NT), 0, st, B, workspace, sums, reg_out, scale);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_noise_normalize(float* const* x, const int32_t* res, int nbufs, float* workspace, void* stream) {
    NoiseBufs B;
    int rc = fill(B, x, nullptr, res, nbufs);
    if (rc) return rc;
    if (workspace == nullptr) {               // one block per buffer, no scratch
        hipLaunchKernelGGL(noise_normalize_kernel, dim3(nbufs), dim3(NT), 0, (hipStream_t)stream, B);
    } else {                                  // workspace: 2 * nbufs floats, zeroed by the caller
        int blocks = 0;
        for (int i = 0; i < nbufs; ++i) blocks += eg3d_cdiv((int64_t)res[i] * res[i], NORM_CHUNK);
        EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, workspace, 2 * nbufs); EG3D_DET_COMMIT(det);
        hipLaunchKernelGGL(noise_moments_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, B, workspace);
        EG3D_DET_END(det);
        hipLaunchKernelGGL(noise_apply_norm_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, B, workspace);
    }
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_adam_step(const eg3d_adam_list* list, float* workspace, void* stream) {
    if (!list || !workspace || list->n < 1 || list->n > EG3D_ADAM_ITEMS_MAX || !list->lr || !list->step) return EG3D_ERR_INVALID;
    int blocks = 0, norm = 0;
    for (int i = 0; i < list->n; ++i) {
        const eg3d_adam_item& it = list->items[i];
        if (!it.p || !it.m || !it.v || (!it.g && !it.g2) || it.n < 1 || it.n > (int64_t)1 << 30) return EG3D_ERR_INVALID;
        blocks += (int)eg3d_cdiv(it.n, ADAM_CHUNK);
        norm |= it.normalize;
    }
    EG3D_DET_SCOPE(det, stream);
    if (norm) { EG3D_DET_BIND(det, workspace, 2 * list->n); }
    EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, *list, workspace);
    EG3D_DET_END(det);
    if (norm) hipLaunchKernelGGL(adam_apply_norm_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, *list, workspace);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_early_stop_flag(const float* value, float threshold, float* done, void* stream) {
    if (!value || !done) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(early_stop_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, value, threshold, done);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
