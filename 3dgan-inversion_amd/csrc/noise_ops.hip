// Noise-buffer maintenance of the latent projector (training/projectors/w_projector.py:221-237, 264-270), fused.
//
// The reference runs, per optimisation step and per noise buffer (13 backbone + 4 SR maps, 4^2 .. 512^2), a pyramid of
// roll / mul / mean / square / avg_pool2d ops, their autograd backward, and a mean/rsqrt renormalisation: ~2500 tiny launches
// per step, which is what bounds the step once the generator is fast.  Here: ONE launch computes the regulariser of all buffers
// AND its gradient (one 1024-thread block per buffer walks the pyramid; levels live in a small scratch), ONE launch renormalises
// all buffers.
//   reg = sum_buffers sum_levels ( mean(x * roll(x,1,W)) ^2 + mean(x * roll(x,1,H)) ^2 ),  levels: res, res/2, ... while res > 8
#include "common.h"

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

__global__ void __launch_bounds__(NT) noise_reg_kernel(const NoiseBufs B, float* __restrict__ ws, float* __restrict__ reg_out, float scale) {
    __shared__ float red[32];
    __shared__ float mx[12], my[12];
    const int b = blockIdx.x;
    const int res0 = B.res[b];
    const float* x0 = B.x[b];
    float* scratch = ws + B.ws_off[b];
    // level pointers: level 0 = input; level l >= 1 at lev_off(l); gradient pyramid after the value pyramid
    int nl = 1;
    for (int r = res0; r > 8; r >>= 1) ++nl;
    float reg = 0.f;
    const float* cur = x0;
    int r = res0;
    int64_t off = 0;
    for (int l = 0; l < nl; ++l) {
        float sx = 0.f, sy = 0.f;
        const int n = r * r;
        const int sh = (r & (r - 1)) == 0 ? __ffs(r) - 1 : -1;          // every StyleGAN resolution is a power of two: shifts instead of a division per element
        for (int i = threadIdx.x; i < n; i += NT) {
            const int y = sh >= 0 ? i >> sh : i / r, xx = i - y * r;
            const float v = cur[i];
            sx += v * cur[y * r + (xx == 0 ? r - 1 : xx - 1)];
            sy += v * cur[(y == 0 ? r - 1 : y - 1) * r + xx];
        }
        sx = block_sum(sx, red) / (float)n;
        sy = block_sum(sy, red) / (float)n;
        if (threadIdx.x == 0) { mx[l] = sx; my[l] = sy; }
        reg += sx * sx + sy * sy;
        if (l + 1 < nl) {
            const int h = r >> 1;
            float* nxt = scratch + off;
            for (int i = threadIdx.x; i < h * h; i += NT) {
                const int y = sh >= 1 ? i >> (sh - 1) : i / h, xx = i - y * h;
                const float* p = cur + (2 * y) * r + 2 * xx;
                nxt[i] = ((p[0] + p[1]) + (p[r] + p[r + 1])) * 0.25f;
            }
            __threadfence_block();
            __syncthreads();
            cur = nxt;
            off += (int64_t)h * h;
            r = h;
        }
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(reg_out, reg * scale);
    float* gout = B.g[b];
    if (gout == nullptr) return;
    __syncthreads();
    // gradient, coarse -> fine.  value level l >= 1 sits at voff[l]; gradient level l >= 1 at vtot + voff[l]
    int64_t voff[12];
    int64_t vtot = 0;
    voff[0] = 0;
    for (int l = 1; l < nl; ++l) { voff[l] = vtot; const int rr = res0 >> l; vtot += (int64_t)rr * rr; }
    for (int l = nl - 1; l >= 0; --l) {
        const int rl = res0 >> l;
        const float* xl = l == 0 ? x0 : scratch + voff[l];
        float* gl = l == 0 ? gout : scratch + vtot + voff[l];
        const float* gup = (l + 1 < nl) ? scratch + vtot + voff[l + 1] : nullptr;
        const float n = (float)(rl * rl);
        const float cxm = 2.f * mx[l] / n * scale, cym = 2.f * my[l] / n * scale;
        const int h = rl >> 1;
        const int shl = (rl & (rl - 1)) == 0 ? __ffs(rl) - 1 : -1;
        for (int i = threadIdx.x; i < rl * rl; i += NT) {
            const int y = shl >= 0 ? i >> shl : i / rl, xx = i - y * rl;
            float g = cxm * (xl[y * rl + (xx == 0 ? rl - 1 : xx - 1)] + xl[y * rl + (xx == rl - 1 ? 0 : xx + 1)]) +
                      cym * (xl[(y == 0 ? rl - 1 : y - 1) * rl + xx] + xl[(y == rl - 1 ? 0 : y + 1) * rl + xx]);
            if (gup) g += gup[(y >> 1) * h + (xx >> 1)] * 0.25f;
            gl[i] = g;
        }
        __threadfence_block();
        __syncthreads();
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
    if (threadIdx.x == 0) { unsafeAtomicAdd(ws + 2 * buf, s); unsafeAtomicAdd(ws + 2 * buf + 1, q); }
}

__global__ void __launch_bounds__(NT) noise_apply_norm_kernel(const NoiseBufs B, const float* __restrict__ ws) {
    int buf, start, n;
    if (!norm_locate(B, blockIdx.x, buf, start, n)) return;
    float* x = B.x[buf];
    const float mean = ws[2 * buf] / (float)n;
    const float inv = 1.0f / sqrtf(ws[2 * buf + 1] / (float)n - mean * mean);
    for (int i = start + threadIdx.x; i < min(n, start + NORM_CHUNK); i += NT) x[i] = (x[i] - mean) * inv;
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
    for (int i = 0; i < nbufs; ++i) t += (int64_t)res[i] * res[i];
    return t;
}

extern "C" int eg3d_noise_regularizer(float* const* x, float* const* grad, const int32_t* res, int nbufs, float* workspace, float* reg_out, float scale,
                                      void* stream) {
    NoiseBufs B;
    int rc = fill(B, x, grad, res, nbufs);
    if (rc) return rc;
    if (!workspace || !reg_out) return EG3D_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    eg3d_zero_words(reg_out, 1, st);
    hipLaunchKernelGGL(noise_reg_kernel, dim3(nbufs), dim3(NT), 0, st, B, workspace, reg_out, scale);
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
        hipLaunchKernelGGL(noise_moments_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, B, workspace);
        hipLaunchKernelGGL(noise_apply_norm_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, B, workspace);
    }
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
