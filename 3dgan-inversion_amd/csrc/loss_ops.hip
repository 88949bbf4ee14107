// Building blocks of the perceptual-loss networks that score every inversion step:
//   * the VGG16-LPIPS feature distance of the latent projector   (training/projectors/w_projector.py:50-52,112,215-219),
//   * the VGG16 conv3_3 features of the depth-reprojection loss   (training/warping_loss.py:31-37),
//   * LPIPS-AlexNet of the pivotal-tuning loss                    (training/coaches/base_coach.py:48,111-112).
// Their convolutions run on the implicit-GEMM kernel of conv_igemm.hip (bias + ReLU fused in its epilogue); this file holds the two
// memory-bound pieces in between, NHWC fp32, four channels (16 bytes) per thread:
//   max pooling (2x2/2 of VGG, 3x3/2 of AlexNet) with the arg-max kept as one byte per element for a gather-form backward, and
//   the LPIPS feature head:  f[c] = scale[c] * x[c] / (||x||_2 + eps) * mul   per pixel (unit-normalise over channels, multiply by
//   the square root of the learned 1x1 "lin" weight and by 1/sqrt(H*W), so that the plain squared distance of two such vectors is
//   the LPIPS distance), written straight into a slice of the flat feature vector.
#include "common.h"
#include "det.h"

namespace {

constexpr int NT = 256;

__global__ void __launch_bounds__(NT) maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int N, int H,
                                                         int W, int C4, int ld4, int k, int s, int Ho, int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C4);
        int64_t r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 a = make_uchar4(0, 0, 0, 0);
        for (int ky = 0; ky < k; ++ky) {
            const int yy = oy * s + ky;
            if (yy >= H) break;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = ox * s + kx;
                if (xx >= W) break;
                const float4 v = reinterpret_cast<const float4*>(x)[((int64_t)(n * H + yy) * W + xx) * ld4 + c];
                const uint8_t t = (uint8_t)(ky * k + kx);
                // first maximum in scan order wins (the tie rule of torch.nn.functional.max_pool2d); NaN propagates
                if (v.x > m.x || v.x != v.x) { m.x = v.x; a.x = t; }
                if (v.y > m.y || v.y != v.y) { m.y = v.y; a.y = t; }
                if (v.z > m.z || v.z != v.z) { m.z = v.z; a.z = t; }
                if (v.w > m.w || v.w != v.w) { m.w = v.w; a.w = t; }
            }
        }
        reinterpret_cast<float4*>(y)[i] = m;
        if (idx) reinterpret_cast<uchar4*>(idx)[i] = a;
    }
}

// dx[n,y,x,c] = sum over the (at most ceil(k/s)^2) windows covering (y,x) whose arg-max is this pixel.  No atomics.
__global__ void __launch_bounds__(NT) maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx, int N,
                                                         int H, int W, int C4, int ld4, int k, int s, int Ho, int Wo) {
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C4);
        int64_t r = i / C4;
        const int xx = (int)(r % W); r /= W;
        const int yy = (int)(r % H);
        const int n = (int)(r / H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        const int oy0 = max(0, (yy - k + s) / s), oy1 = min(Ho - 1, yy / s);
        const int ox0 = max(0, (xx - k + s) / s), ox1 = min(Wo - 1, xx / s);
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) {
                const uint8_t t = (uint8_t)((yy - oy * s) * k + (xx - ox * s));
                const int64_t o = ((int64_t)(n * Ho + oy) * Wo + ox) * C4 + c;
                const uchar4 a = reinterpret_cast<const uchar4*>(idx)[o];
                const float4 v = reinterpret_cast<const float4*>(dy)[o];
                if (a.x == t) g.x += v.x;
                if (a.y == t) g.y += v.y;
                if (a.z == t) g.z += v.z;
                if (a.w == t) g.w += v.w;
            }
        reinterpret_cast<float4*>(dx)[((int64_t)(n * H + yy) * W + xx) * ld4 + c] = g;
    }
}

__host__ __device__ int group_width(int C4);

// One pixel per group of G = min(64, C/4 rounded up to a power of two) lanes; each lane walks the pixel's channels four at a time.
template <bool BWD>
__device__ __forceinline__ void unit_normalize_body(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ dout,
                                                    float* __restrict__ out, int64_t npix, int HW, int C4, int ld4, float mul, float eps,
                                                    int64_t o_nstride, int G, int eps_inside) {
    const int lane = threadIdx.x % G;
    const int64_t gid = ((int64_t)blockIdx.x * NT + threadIdx.x) / G;
    const int64_t ngroups = (int64_t)gridDim.x * NT / G;
    for (int64_t pix = gid; pix < npix; pix += ngroups) {
        const float4* xp = reinterpret_cast<const float4*>(x) + pix * ld4;
        const int64_t n = pix / HW, q = pix - n * HW;
        // dout / features live in a flat [N, F] vector: this layer's slice starts at the pointer passed in, batch stride o_nstride
        const int64_t fo = n * o_nstride + q * (int64_t)C4 * 4;
        float ss = 0.f, dot = 0.f;
        for (int c = lane; c < C4; c += G) {
            const float4 v = xp[c];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            if (BWD) {
                const float4 g = *reinterpret_cast<const float4*>(dout + fo + 4 * c);
                const float4 s = scale ? reinterpret_cast<const float4*>(scale)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
                dot += v.x * g.x * s.x + v.y * g.y * s.y + v.z * g.z * s.z + v.w * g.w * s.w;
            }
        }
        for (int m = G >> 1; m >= 1; m >>= 1) {
            ss += __shfl_xor(ss, m);
            if (BWD) dot += __shfl_xor(dot, m);
        }
        const float r = sqrtf(ss);
        // n = x/(r+eps):  dx = g/(r+eps) - x (x.g) / (r (r+eps)^2).  An all-zero pixel (r = 0) gets gradient 0: autograd of the reference
        // expression yields NaN there (0 * inf through sqrt), the limit g/eps is 1e10 * g -- neither is useful to an optimiser.
        float inv = (BWD && r == 0.f) ? 0.f : mul / (r + eps);
        float back = (BWD && r > 0.f) ? dot * mul / (r * (r + eps) * (r + eps)) : 0.f;
        if (eps_inside) {       // n = x rsqrt(|x|^2 + eps):  dx = g q - x (x.g) q^3,  q = (|x|^2 + eps)^-1/2  (smooth at 0)
            const float q = 1.0f / sqrtf(ss + eps);
            inv = mul * q;
            back = BWD ? dot * mul * q * q * q : 0.f;
        }
        for (int c = lane; c < C4; c += G) {
            const float4 v = xp[c];
            const float4 s = scale ? reinterpret_cast<const float4*>(scale)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
            float4 o;
            if (!BWD) {
                o = make_float4(v.x * s.x * inv, v.y * s.y * inv, v.z * s.z * inv, v.w * s.w * inv);
                *reinterpret_cast<float4*>(out + fo + 4 * c) = o;
            } else {
                const float4 g = *reinterpret_cast<const float4*>(dout + fo + 4 * c);
                o = make_float4(g.x * s.x * inv - v.x * back, g.y * s.y * inv - v.y * back, g.z * s.z * inv - v.z * back,
                                g.w * s.w * inv - v.w * back);
                reinterpret_cast<float4*>(out)[pix * ld4 + c] = o;
            }
        }
    }
}

template <bool BWD>
__global__ void __launch_bounds__(NT) unit_normalize_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ dout,
                                                            float* __restrict__ out, int64_t npix, int HW, int C4, int ld4, float mul, float eps,
                                                            int64_t o_nstride, int G, int eps_inside) {
    unit_normalize_body<BWD>(x, scale, dout, out, npix, HW, C4, ld4, mul, eps, o_nstride, G, eps_inside);
}

// Every tap of the feature pyramid in one launch: blockIdx.y = level (the grid-stride loop inside absorbs the different pixel counts).
template <bool BWD>
__global__ void __launch_bounds__(NT) unit_normalize_levels_kernel(const eg3d_unit_levels b) {
    const eg3d_unit_level& l = b.levels[blockIdx.y];
    unit_normalize_body<BWD>(l.x, l.scale, BWD ? l.feat : nullptr, BWD ? l.dx : const_cast<float*>(l.feat), (int64_t)b.N * l.HW, l.HW, l.C / 4,
                             l.ldx / 4, l.mul, b.eps, b.feat_nstride, group_width(l.C / 4), b.eps_inside);
}

// Generator image -> feature-net input (w_projector.py:198-200,215: (img + 1) * 255/2, area-resized to 256^2): one pass instead of
// slice / scale / shift / pool / pad / layout copies.  Thread = output pixel; img pixels are 16 bytes (3 used channels + padding).
__global__ void __launch_bounds__(NT) image_prepare_fwd_kernel(const float4* __restrict__ img, float4* __restrict__ out, int N, int Ho, int Wo, int f,
                                                               float mul, float add) {
    const int64_t total = (int64_t)N * Ho * Wo;
    const int W = Wo * f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int ox = (int)(i % Wo);
        const int64_t r = i / Wo;                  // n * Ho + oy
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int ky = 0; ky < f; ++ky)
            for (int kx = 0; kx < f; ++kx) {
                const float4 v = img[(r * f + ky) * W + ox * f + kx];
                sx += v.x; sy += v.y; sz += v.z;
            }
        const float m = mul / (float)(f * f);
        out[i] = make_float4(fmaf(sx, m, add), fmaf(sy, m, add), fmaf(sz, m, add), 0.f);
    }
}

__global__ void __launch_bounds__(NT) image_prepare_bwd_kernel(const float4* __restrict__ dout, float4* __restrict__ dimg, int N, int H, int W, int f, float mul) {
    const int64_t total = (int64_t)N * H * W;
    const int Wo = W / f;
    const float m = mul / (float)(f * f);
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W);
        const int64_t r = i / W;                   // n * H + y
        const int64_t ro = (r / H) * (H / f) + (r % H) / f;
        const float4 g = dout[ro * Wo + x / f];
        dimg[i] = make_float4(g.x * m, g.y * m, g.z * m, 0.f);
    }
}

// Squared distance of two flat feature vectors per image and its gradient (the projector's `dist`, w_projector.py:216-219).
__global__ void __launch_bounds__(NT) sqdist_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t F) {
    const int n = blockIdx.y;
    const float4* a4 = reinterpret_cast<const float4*>(a + n * F);
    const float4* b4 = reinterpret_cast<const float4*>(b + n * F);
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < F / 4; i += (int64_t)gridDim.x * NT) {
        const float4 u = a4[i], v = b4[i];
        const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
        s += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    __shared__ float red[NT / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) eg3d_acc(out + n, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void __launch_bounds__(NT) sqdist_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                                        float* __restrict__ da, int64_t F, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total4; i += (int64_t)gridDim.x * NT) {
        const float k = 2.f * g[(i * 4) / F];
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(da)[i] = make_float4(k * (u.x - v.x), k * (u.y - v.y), k * (u.z - v.z), k * (u.w - v.w));
    }
}

// The pivotal-tuning objective (base_coach.py:104-126 calc_loss: L2 + LPIPS at both resolutions, plus the depth total variation of
// :294-305) as weighted sums of a few reductions: every term kernel adds its own value to `term` and its weighted share to `total`,
// so the objective needs no scalar glue launches; the backward kernels take the incoming scalar gradient by pointer and the term's
// weight by value.
__device__ __forceinline__ void block_sum_commit(float s, float* __restrict__ term, float term_scale, float* __restrict__ total, float total_scale) {
    __shared__ float red[NT / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (red[0] + red[1]) + (red[2] + red[3]);
        if (term) eg3d_acc(term, v * term_scale);
        if (total) eg3d_acc(total, v * total_scale);
    }
}

__global__ void __launch_bounds__(NT) sqdist_sum_fwd_kernel(const float4* __restrict__ a, const float4* __restrict__ b, int64_t total4, float* __restrict__ term,
                                                            float term_scale, float* __restrict__ total, float total_scale) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total4; i += (int64_t)gridDim.x * NT) {
        const float4 u = a[i], v = b[i];
        const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
        s += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    block_sum_commit(s, term, term_scale, total, total_scale);
}

__global__ void __launch_bounds__(NT) sqdist_sum_bwd_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float* __restrict__ g, float gscale,
                                                            float4* __restrict__ da, int64_t total4) {
    const float k = 2.f * gscale * g[0];
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total4; i += (int64_t)gridDim.x * NT) {
        const float4 u = a[i], v = b[i];
        da[i] = make_float4(k * (u.x - v.x), k * (u.y - v.y), k * (u.z - v.z), k * (u.w - v.w));
    }
}

// Squared forward-difference total variation of a [B,H,W] map: sum over y < H-1, x < W-1 of (v - v_right)^2 + (v - v_below)^2.
__global__ void __launch_bounds__(NT) tv_norm_fwd_kernel(const float* __restrict__ v, int B, int H, int W, float* __restrict__ term, float term_scale,
                                                         float* __restrict__ total, float total_scale) {
    const int64_t n = (int64_t)B * H * W;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        if (x < W - 1 && y < H - 1) {
            const float c = v[i], r = c - v[i + 1], d = c - v[i + W];
            s += r * r + d * d;
        }
    }
    block_sum_commit(s, term, term_scale, total, total_scale);
}

__global__ void __launch_bounds__(NT) tv_norm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float gscale, float* __restrict__ dv, int B,
                                                         int H, int W) {
    const int64_t n = (int64_t)B * H * W;
    const float k = 2.f * gscale * g[0];
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float c = v[i];
        float a = 0.f;
        if (x < W - 1 && y < H - 1) a += (c - v[i + 1]) + (c - v[i + W]);       // as the centre of its own pair of differences
        if (x >= 1 && y < H - 1) a -= v[i - 1] - c;                             // as the right neighbour of (y, x-1)
        if (y >= 1 && x < W - 1) a -= v[i - W] - c;                             // as the lower neighbour of (y-1, x)
        dv[i] = k * a;
    }
}

// The raw (neural-rendering resolution) RGB image with 4-float pixels from the C-channel rendered feature image (triplane.py:84-85:
// rgb = features[:, :3]): y[p] = (x[p,0], x[p,1], x[p,2], 0); backward scatters into a zero-filled [P,C] gradient.
__global__ void __launch_bounds__(NT) slice_rgb4_fwd_kernel(const float* __restrict__ x, float4* __restrict__ y, int64_t P, int C) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P) return;
    const float4 v = *reinterpret_cast<const float4*>(x + i * C);
    y[i] = make_float4(v.x, v.y, v.z, 0.f);
}

__global__ void __launch_bounds__(NT) slice_rgb4_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ addend, float4* __restrict__ dx, int64_t P, int C4) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P * C4) return;
    float4 o = addend != nullptr ? addend[i] : make_float4(0.f, 0.f, 0.f, 0.f);       // (the gradient the feature image's other consumer, the SR head, sent)
    if (i % C4 == 0) { const float4 g = dy[i / C4]; o.x += g.x; o.y += g.y; o.z += g.z; }
    dx[i] = o;
}

// Depth-reprojection geometry of the warping loss (training/warping_loss.py:18-54, LinePlaneCollision :58-72): lift a pixel with its
// rendered depth along its ray, intersect the line from the canonical camera centre through that point with the canonical image plane,
// project with the canonical world->camera matrix and the intrinsics, map to [-1,1].  One thread per pixel; constants [24]:
// c[3] canonical camera centre, P0[3] point on its image plane, A[9] + b[3] rows 0..2 of the world->camera matrix, K[6] rows 0..1 of the intrinsics.
struct WarpPix { float ux, uy, uz, ndotu, si, qx, qy, qz; };
__device__ __forceinline__ WarpPix warp_pixel(const float* __restrict__ k, float ox, float oy, float oz, float dx, float dy, float dz, float t) {
    WarpPix w;
    w.ux = ox + dx * t - k[0]; w.uy = oy + dy * t - k[1]; w.uz = oz + dz * t - k[2];        // u = xyz - c
    const float nx = -k[0], ny = -k[1], nz = -k[2];                                         // plane normal = -c
    w.ndotu = nx * w.ux + ny * w.uy + nz * w.uz;
    const float kk = -(nx * (k[0] - k[3]) + ny * (k[1] - k[4]) + nz * (k[2] - k[5]));       // -(n . (c - P0))
    w.si = kk / w.ndotu;
    const float hx = k[0] + w.si * w.ux, hy = k[1] + w.si * w.uy, hz = k[2] + w.si * w.uz;
    w.qx = k[6] * hx + k[7] * hy + k[8] * hz + k[15];
    w.qy = k[9] * hx + k[10] * hy + k[11] * hz + k[16];
    w.qz = k[12] * hx + k[13] * hy + k[14] * hz + k[17];
    return w;
}

__global__ void __launch_bounds__(NT) warp_project_fwd_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ depth,
                                                              const float* __restrict__ k, float2* __restrict__ uv, int64_t P) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P) return;
    const WarpPix w = warp_pixel(k, o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2], depth[i]);
    const float rx = w.qx / w.qz, ry = w.qy / w.qz;
    uv[i] = make_float2((k[18] * rx + k[19] * ry + k[20] - 0.5f) * 2.f, (k[21] * rx + k[22] * ry + k[23] - 0.5f) * 2.f);
}

__global__ void __launch_bounds__(NT) warp_project_bwd_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ depth,
                                                              const float* __restrict__ k, const float2* __restrict__ duv, float* __restrict__ d_o,
                                                              float* __restrict__ d_d, float* __restrict__ d_depth, int64_t P) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P) return;
    const float dx = d[3 * i], dy = d[3 * i + 1], dz = d[3 * i + 2], t = depth[i];
    const WarpPix w = warp_pixel(k, o[3 * i], o[3 * i + 1], o[3 * i + 2], dx, dy, dz, t);
    const float2 g = duv[i];
    const float drx = 2.f * (k[18] * g.x + k[21] * g.y), dry = 2.f * (k[19] * g.x + k[22] * g.y);
    const float iz = 1.f / w.qz;
    const float dqx = drx * iz, dqy = dry * iz, dqz = -(drx * w.qx + dry * w.qy) * iz * iz;
    const float dhx = k[6] * dqx + k[9] * dqy + k[12] * dqz, dhy = k[7] * dqx + k[10] * dqy + k[13] * dqz, dhz = k[8] * dqx + k[11] * dqy + k[14] * dqz;
    const float dsi = dhx * w.ux + dhy * w.uy + dhz * w.uz;
    const float dn = -w.si / w.ndotu * dsi;                                                  // si = kk / ndotu
    const float gx = w.si * dhx - k[0] * dn, gy = w.si * dhy - k[1] * dn, gz = w.si * dhz - k[2] * dn;     // d u = si d hit + n d ndotu,  n = -c
    d_o[3 * i] = gx; d_o[3 * i + 1] = gy; d_o[3 * i + 2] = gz;
    d_d[3 * i] = gx * t; d_d[3 * i + 1] = gy * t; d_d[3 * i + 2] = gz * t;
    d_depth[i] = gx * dx + gy * dy + gz * dz;
}

// F.grid_sample(input, grid, mode='bilinear', padding_mode='zeros', align_corners=False) on a channels-last input (the feature warp of the
// depth-reprojection loss, training/warping_loss.py:50): input [N,H,W,C], grid [N,Ho,Wo,2] in [-1,1], out [N,Ho,Wo,C]; C % 4 == 0.
// One wave per output pixel: lane = channel quad (loops if C > 256), the four corner rows are contiguous C-float runs.  ATen's kernel walks the
// channels of an NCHW-indexed tensor one by one per thread (152 us forward / 218 us backward for 256 x 64^2 on a channels-last map).
__device__ __forceinline__ void gs_corners(float gx, float gy, int H, int W, int& x0, int& y0, float& tx, float& ty) {
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    x0 = (int)fx; y0 = (int)fy; tx = ix - fx; ty = iy - fy;
}

__global__ void __launch_bounds__(256) grid_sample_nhwc_fwd_kernel(const float* __restrict__ inp, const float2* __restrict__ grid, float* __restrict__ out,
                                                                  int64_t P, int64_t ppi, int H, int W, int C) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const float2 g = grid[p];
    int x0, y0; float tx, ty;
    gs_corners(g.x, g.y, H, W, x0, y0, tx, ty);
    const float* base = inp + (p / ppi) * (int64_t)H * W * C;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    const bool xa = (unsigned)x0 < (unsigned)W, xb = (unsigned)(x0 + 1) < (unsigned)W, ya = (unsigned)y0 < (unsigned)H, yb = (unsigned)(y0 + 1) < (unsigned)H;
    for (int c = lane * 4; c < C; c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add = [&](bool ok, int yy, int xx, float w) {
            if (!ok) return;
            const float4 t = *reinterpret_cast<const float4*>(base + ((int64_t)yy * W + xx) * C + c);
            acc.x = fmaf(w, t.x, acc.x); acc.y = fmaf(w, t.y, acc.y); acc.z = fmaf(w, t.z, acc.z); acc.w = fmaf(w, t.w, acc.w);
        };
        add(xa && ya, y0, x0, w00); add(xb && ya, y0, x0 + 1, w01); add(xa && yb, y0 + 1, x0, w10); add(xb && yb, y0 + 1, x0 + 1, w11);
        *reinterpret_cast<float4*>(out + p * C + c) = acc;
    }
}

// d grid (always) and d input (optional, pre-zeroed, accumulated) from d out
__global__ void __launch_bounds__(256) grid_sample_nhwc_bwd_kernel(const float* __restrict__ inp, const float2* __restrict__ grid, const float* __restrict__ dout,
                                                                  float2* __restrict__ dgrid, float* __restrict__ dinp, int64_t P, int64_t ppi, int H, int W, int C) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63;
    const float2 g = grid[p];
    int x0, y0; float tx, ty;
    gs_corners(g.x, g.y, H, W, x0, y0, tx, ty);
    const int64_t ioff = (p / ppi) * (int64_t)H * W * C;
    const bool xa = (unsigned)x0 < (unsigned)W, xb = (unsigned)(x0 + 1) < (unsigned)W, ya = (unsigned)y0 < (unsigned)H, yb = (unsigned)(y0 + 1) < (unsigned)H;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    float gx = 0.f, gy = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 d = *reinterpret_cast<const float4*>(dout + p * C + c);
        auto corner = [&](bool ok, int yy, int xx, float w, float sx, float sy) {       // value weight w; d w / d ix = sx, d w / d iy = sy
            if (!ok) return;
            const int64_t o = ioff + ((int64_t)yy * W + xx) * C + c;
            const float4 t = *reinterpret_cast<const float4*>(inp + o);
            const float dot = d.x * t.x + d.y * t.y + d.z * t.z + d.w * t.w;
            gx = fmaf(sx, dot, gx); gy = fmaf(sy, dot, gy);
            if (dinp != nullptr) { eg3d_acc(dinp + o, w * d.x); eg3d_acc(dinp + o + 1, w * d.y); eg3d_acc(dinp + o + 2, w * d.z); eg3d_acc(dinp + o + 3, w * d.w); }
        };
        corner(xa && ya, y0, x0, w00, -(1.f - ty), -(1.f - tx));
        corner(xb && ya, y0, x0 + 1, w01, (1.f - ty), -tx);
        corner(xa && yb, y0 + 1, x0, w10, -ty, (1.f - tx));
        corner(xb && yb, y0 + 1, x0 + 1, w11, ty, tx);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); }
    if (lane == 0) dgrid[p] = make_float2(gx * (float)W * 0.5f, gy * (float)H * 0.5f);
}

int grid_blocks(int64_t threads) {
    const int64_t b = (threads + NT - 1) / NT;
    return (int)(b < 8192 ? b : 8192);
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int pool_check(const void* a, const void* b, int N, int H, int W, int C, int ld, int k, int s) {
    if (!a || !b || N < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || ld < C || (ld & 3) || k < 1 || k > 15 || s < 1 || H < k || W < k) return EG3D_ERR_INVALID;
    if (!aligned16(a) || !aligned16(b)) return EG3D_ERR_INVALID;
    return EG3D_OK;
}

__host__ __device__ int group_width(int C4) {
    int g = 1;
    while (g < C4 && g < 64) g <<= 1;
    return g;
}


// ---- small-channel 3 x 3 convolutions on the vector ALUs (the stand-in feature pyramid: 4 -> 16 -> 32 -> 64 channels at 256^2 .. 64^2) ---------
// Those layers are 0.08 - 0.15 GFLOP: on the implicit-GEMM kernel each is a 17 - 23 us launch (a 128 x 32 tile walks a 36 .. 288-deep
// contraction in barrier-separated steps) plus separate pooling / activation-backward / gradient-add passes.  Here a thread owns a 2 x 2 quad
// of output pixels and G output channels, walks the input channels four at a time (a 4 x 4 patch of 16-byte pixels in registers per step) and
// multiplies in exact fp32; the weights of a workgroup's channel group are wave-uniform (scalar loads, SGPR operands of v_fmac).  The
// epilogue applies lrelu * gain and (POOL) writes the 2 x 2 average next to the full-resolution tensor the backward needs for the sign.
// The data gradient is the same kernel on the flipped, transposed weights.
template <int G, bool POOL, int KS, bool XF = false>         // XF: the input is formed on the fly (see eg3d_conv3x3_direct_params::ga).  KS: the input-channel quads are dealt to KS waves of the block (64 output quads per block), partial sums meet in LDS
__global__ void __launch_bounds__(64 * KS) conv3x3_direct_kernel(const eg3d_conv3x3_direct_params p) {
    __shared__ float part[KS > 1 ? (KS - 1) * 64 * 4 * G : 1];
    const int Hq = p.H >> 1, Wq = p.W >> 1;
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const bool live = q < Hq * Wq;
    const int qy = live ? q / Wq : 0, qx = live ? q - qy * Wq : 0;
    const int n = blockIdx.z, cog = blockIdx.y;
    const int y0 = 2 * qy - 1, x0 = 2 * qx - 1;                      // top-left corner of the 4 x 4 input patch
    const float* xn = p.x + (int64_t)n * p.H * p.W * p.Ci;
    const float* wq = p.w + (int64_t)cog * (p.Ci >> 2) * 9 * 4 * G;
    float acc[4][G];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[i][g] = 0.f;
    // patch addresses / validity do not depend on the channel step
    int off[16];
    bool ok[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int yy = y0 + (i >> 2), xx = x0 + (i & 3);
        ok[i] = live && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        off[i] = ok[i] ? (yy * p.W + xx) * p.Ci : 0;
    }
    // XF: the 4 x 4 patch spans the 3 x 3 pooled cells (qy - 1 .. qy + 1, qx - 1 .. qx + 1); patch row / column i lies in cell (i + 1) >> 1
    int goff[9];
    if constexpr (XF) {
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int cy = min(max(qy - 1 + c / 3, 0), Hq - 1), cx = min(max(qx - 1 + c % 3, 0), Wq - 1);      // (cells outside the image are only met by patch pixels with ok = false)
            goff[c] = ((n * Hq + cy) * Wq + cx) * p.Ci;
        }
    }
    const float xf_pos = 0.25f * p.gain, xf_neg = 0.25f * p.gain * p.alpha;
    for (int cq = ks; cq < (p.Ci >> 2); cq += KS) {
        float4 t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ok[i] ? *reinterpret_cast<const float4*>(xn + off[i] + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (XF) {             // x is the saved activation output: dz = 0.25 (ga + gb)[cell] * gain * (y > 0 ? 1 : alpha)   (eg3d_pool2_act_bwd)
            float4 gsum[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                gsum[c] = p.ga != nullptr ? *reinterpret_cast<const float4*>(p.ga + goff[c] + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.gb != nullptr) { const float4 h = *reinterpret_cast<const float4*>(p.gb + goff[c] + 4 * cq); gsum[c].x += h.x; gsum[c].y += h.y; gsum[c].z += h.z; gsum[c].w += h.w; }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 g = gsum[(((i >> 2) + 1) >> 1) * 3 + (((i & 3) + 1) >> 1)], yv = t[i];
                t[i] = ok[i] ? make_float4(g.x * (yv.x > 0.f ? xf_pos : xf_neg), g.y * (yv.y > 0.f ? xf_pos : xf_neg), g.z * (yv.z > 0.f ? xf_pos : xf_neg),
                                           g.w * (yv.w > 0.f ? xf_pos : xf_neg)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float* wc = wq + (int64_t)cq * 9 * 4 * G;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float wv = wc[(tap * 4 + j) * G + g];
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const float4 tv = t[((o >> 1) + ky) * 4 + (o & 1) + kx];
                        const float xv = j == 0 ? tv.x : (j == 1 ? tv.y : (j == 2 ? tv.z : tv.w));
                        acc[o][g] = fmaf(xv, wv, acc[o][g]);
                    }
                }
            }
        }
    }
    if constexpr (KS > 1) {             // slices 1 .. KS-1 hand their partial sums to slice 0 (lane-major: conflict-free)
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) part[((ks - 1) * 4 * G + i * G + g) * 64 + lane] = acc[i][g];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 0; k < KS - 1; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[i][g] += part[(k * 4 * G + i * G + g) * 64 + lane];
    }
    if (!live) return;
    const float gpos = p.gain, gneg = p.gain * p.alpha;
    float pool[G];
#pragma unroll
    for (int g = 0; g < G; ++g) pool[g] = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            v[g] = p.act ? acc[o][g] * (acc[o][g] > 0.f ? gpos : gneg) : acc[o][g];
            pool[g] += v[g];
        }
        if (p.y != nullptr) {
            float* yo = p.y + (((int64_t)n * p.H + 2 * qy + (o >> 1)) * p.W + 2 * qx + (o & 1)) * p.Co + cog * G;
            if constexpr (G == 4) *reinterpret_cast<float4*>(yo) = make_float4(v[0], v[1], v[2], v[3]);
            else if constexpr (G == 2) *reinterpret_cast<float2*>(yo) = make_float2(v[0], v[1]);
            else yo[0] = v[0];
        }
    }
    if constexpr (POOL) {
        float* po = p.pooled + (((int64_t)n * Hq + qy) * Wq + qx) * p.Co + cog * G;
        if constexpr (G == 4) *reinterpret_cast<float4*>(po) = make_float4(pool[0] * 0.25f, pool[1] * 0.25f, pool[2] * 0.25f, pool[3] * 0.25f);
        else if constexpr (G == 2) *reinterpret_cast<float2*>(po) = make_float2(pool[0] * 0.25f, pool[1] * 0.25f);
        else po[0] = pool[0] * 0.25f;
    }
}

// dz[n,y,x,c] = 0.25 (ga + gb)[n,y/2,x/2,c] * gain * (yref[n,y,x,c] > 0 ? 1 : alpha): the 2 x 2 average's backward, the sum of the pooled
// tensor's two consumers' gradients and the lrelu backward (bias_act.cu:76,145 semantics on the saved output) in one pass
__global__ void __launch_bounds__(NT) pool2_act_bwd_kernel(const float4* __restrict__ ga, const float4* __restrict__ gb, const float4* __restrict__ yref,
                                                           float4* __restrict__ dz, int N, int H, int W, int C4, float alpha, float gain) {
    const int64_t total = (int64_t)N * H * W * C4;
    const int Hq = H >> 1, Wq = W >> 1;
    const float gp = 0.25f * gain, gn = 0.25f * gain * alpha;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += (int64_t)gridDim.x * NT) {
        const int c = (int)(i % C4);
        int64_t r = i / C4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int n = (int)(r / H);
        const int64_t j = (((int64_t)n * Hq + (y >> 1)) * Wq + (x >> 1)) * C4 + c;
        float4 g = ga != nullptr ? ga[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (gb != nullptr) { const float4 h = gb[j]; g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        const float4 yv = yref[i];
        dz[i] = make_float4(g.x * (yv.x > 0.f ? gp : gn), g.y * (yv.y > 0.f ? gp : gn), g.z * (yv.z > 0.f ? gp : gn), g.w * (yv.w > 0.f ? gp : gn));
    }
}

}  // namespace

extern "C" int eg3d_conv3x3_direct(const eg3d_conv3x3_direct_params* p, void* stream) {
    if (!p || !p->x || !p->w || (!p->y && !p->pooled)) return EG3D_ERR_INVALID;
    if (p->N < 1 || p->H < 2 || p->W < 2 || (p->H & 1) || (p->W & 1) || p->Ci < 4 || (p->Ci & 3) || p->Co < 1) return EG3D_ERR_UNSUPPORTED;
    if ((p->G != 1 && p->G != 2 && p->G != 4) || p->Co % p->G || (int64_t)p->H * p->W * p->Ci > INT32_MAX || p->N > 65535 || p->Co / p->G > 65535) return EG3D_ERR_UNSUPPORTED;
    if (!aligned16(p->x) || !aligned16(p->w) || (p->y && !aligned16(p->y)) || (p->pooled && !aligned16(p->pooled)) || (p->G > 1 && (p->Co * 4) % (4 * p->G))) return EG3D_ERR_INVALID;
    const dim3 grid((unsigned)(((int64_t)(p->H / 2) * (p->W / 2) + 63) / 64), (unsigned)(p->Co / p->G), (unsigned)p->N);
    const bool pool = p->pooled != nullptr;
    const bool xf = p->ga != nullptr || p->gb != nullptr;
    if (xf && (p->act || pool || (p->ga && !aligned16(p->ga)) || (p->gb && !aligned16(p->gb)))) return EG3D_ERR_INVALID;      // alpha / gain belong to the input transform then
    const int ks = p->Ci >= 16 ? 4 : 1;            // four waves share a block's contraction once it is four channel quads deep
#define EG3D_C3D2(G_, KS_) do { if (pool) hipLaunchKernelGGL((conv3x3_direct_kernel<G_, true, KS_>), grid, dim3(64 * KS_), 0, (hipStream_t)stream, *p); \
                               else if (xf) hipLaunchKernelGGL((conv3x3_direct_kernel<G_, false, KS_, true>), grid, dim3(64 * KS_), 0, (hipStream_t)stream, *p); \
                               else hipLaunchKernelGGL((conv3x3_direct_kernel<G_, false, KS_>), grid, dim3(64 * KS_), 0, (hipStream_t)stream, *p); } while (0)
#define EG3D_C3D(G_) do { if (ks == 4) EG3D_C3D2(G_, 4); else EG3D_C3D2(G_, 1); } while (0)
    if (p->G == 4) EG3D_C3D(4); else if (p->G == 2) EG3D_C3D(2); else EG3D_C3D(1);
#undef EG3D_C3D2
#undef EG3D_C3D
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_pool2_act_bwd(const float* ga, const float* gb, const float* yref, float* dz, int N, int H, int W, int C, float alpha, float gain, void* stream) {
    if (!yref || !dz || (!ga && !gb) || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || (C & 3)) return EG3D_ERR_INVALID;
    if ((ga && !aligned16(ga)) || (gb && !aligned16(gb)) || !aligned16(yref) || !aligned16(dz)) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(pool2_act_bwd_kernel, dim3(grid_blocks((int64_t)N * H * W * (C / 4))), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(ga ? ga : gb), reinterpret_cast<const float4*>(ga ? gb : nullptr), reinterpret_cast<const float4*>(yref),
                       reinterpret_cast<float4*>(dz), N, H, W, C / 4, alpha, gain);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}


extern "C" int eg3d_maxpool2d_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, int ldx, int k, int s, void* stream) {
    int rc = pool_check(x, y, N, H, W, C, ldx, k, s);
    if (rc) return rc;
    const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    const int blocks = grid_blocks(total);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, y, argmax, N, H, W, C / 4, ldx / 4, k, s, Ho, Wo);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_maxpool2d_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int H, int W, int C, int ldx, int k, int s, void* stream) {
    int rc = pool_check(dy, dx, N, H, W, C, ldx, k, s);
    if (rc) return rc;
    if (!argmax) return EG3D_ERR_INVALID;
    const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    const int blocks = grid_blocks(total);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, dy, argmax, dx, N, H, W, C / 4, ldx / 4, k, s, Ho, Wo);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_unit_normalize_fwd(const float* x, const float* scale, float* feat, int N, int HW, int C, int ldx, float mul, float eps,
                                       int64_t feat_nstride, int eps_inside, void* stream) {
    if (!x || !feat || N < 1 || HW < 1 || C < 4 || (C & 3) || ldx < C || (ldx & 3) || !aligned16(x) || !aligned16(feat) || (feat_nstride & 3) ||
        (scale && !aligned16(scale)))
        return EG3D_ERR_INVALID;
    const int G = group_width(C / 4);
    const int64_t npix = (int64_t)N * HW;
    const int blocks = grid_blocks(npix * G);
    hipLaunchKernelGGL(unit_normalize_kernel<false>, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, scale, (const float*)nullptr, feat, npix, HW,
                       C / 4, ldx / 4, mul, eps, feat_nstride, G, eps_inside);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_unit_normalize_bwd(const float* x, const float* scale, const float* dfeat, float* dx, int N, int HW, int C, int ldx, float mul,
                                       float eps, int64_t feat_nstride, int eps_inside, void* stream) {
    if (!x || !dfeat || !dx || N < 1 || HW < 1 || C < 4 || (C & 3) || ldx < C || (ldx & 3) || !aligned16(x) || !aligned16(dfeat) || !aligned16(dx) ||
        (feat_nstride & 3) || (scale && !aligned16(scale)))
        return EG3D_ERR_INVALID;
    const int G = group_width(C / 4);
    const int64_t npix = (int64_t)N * HW;
    const int blocks = grid_blocks(npix * G);
    hipLaunchKernelGGL(unit_normalize_kernel<true>, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, scale, dfeat, dx, npix, HW, C / 4, ldx / 4, mul,
                       eps, feat_nstride, G, eps_inside);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_unit_normalize_levels(const eg3d_unit_levels* batch, int bwd, void* stream) {
    if (!batch || batch->n < 1 || batch->n > EG3D_UNIT_LEVELS_MAX || batch->N < 1 || (batch->feat_nstride & 3)) return EG3D_ERR_INVALID;
    int64_t most = 0;
    for (int i = 0; i < batch->n; ++i) {
        const eg3d_unit_level& l = batch->levels[i];
        if (!l.x || !l.feat || (bwd && !l.dx) || l.HW < 1 || l.C < 4 || (l.C & 3) || l.ldx < l.C || (l.ldx & 3) || !aligned16(l.x) ||
            !aligned16(l.feat) || (bwd && !aligned16(l.dx)) || (l.scale && !aligned16(l.scale)))
            return EG3D_ERR_INVALID;
        most = std::max<int64_t>(most, (int64_t)batch->N * l.HW * group_width(l.C / 4));
    }
    const dim3 grid(grid_blocks(most), batch->n);
    if (bwd) hipLaunchKernelGGL(unit_normalize_levels_kernel<true>, grid, dim3(NT), 0, (hipStream_t)stream, *batch);
    else hipLaunchKernelGGL(unit_normalize_levels_kernel<false>, grid, dim3(NT), 0, (hipStream_t)stream, *batch);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_image_prepare_fwd(const float* img, float* out, int N, int H, int W, int factor, float mul, float add, void* stream) {
    if (!img || !out || N < 1 || H < 1 || W < 1 || factor < 1 || H % factor || W % factor || !aligned16(img) || !aligned16(out)) return EG3D_ERR_INVALID;
    const int Ho = H / factor, Wo = W / factor;
    hipLaunchKernelGGL(image_prepare_fwd_kernel, dim3(grid_blocks((int64_t)N * Ho * Wo)), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(img), reinterpret_cast<float4*>(out), N, Ho, Wo, factor, mul, add);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_image_prepare_bwd(const float* dout, float* dimg, int N, int H, int W, int factor, float mul, void* stream) {
    if (!dout || !dimg || N < 1 || H < 1 || W < 1 || factor < 1 || H % factor || W % factor || !aligned16(dout) || !aligned16(dimg)) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(image_prepare_bwd_kernel, dim3(grid_blocks((int64_t)N * H * W)), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(dout), reinterpret_cast<float4*>(dimg), N, H, W, factor, mul);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_sqdist_fwd(const float* a, const float* b, float* out, int N, int64_t F, void* stream) {
    if (!a || !b || !out || N < 1 || F < 4 || (F & 3) || !aligned16(a) || !aligned16(b)) return EG3D_ERR_INVALID;
    const int bx = (int)std::min<int64_t>(256, (F / 4 + NT * 4 - 1) / (NT * 4));
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, out, N); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(sqdist_fwd_kernel, dim3(std::max(bx, 1), N), dim3(NT), 0, (hipStream_t)stream, a, b, out, F);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_sqdist_bwd(const float* a, const float* b, const float* g, float* da, int N, int64_t F, void* stream) {
    if (!a || !b || !g || !da || N < 1 || F < 4 || (F & 3) || !aligned16(a) || !aligned16(b) || !aligned16(da)) return EG3D_ERR_INVALID;
    const int64_t total4 = (int64_t)N * F / 4;
    hipLaunchKernelGGL(sqdist_bwd_kernel, dim3(grid_blocks(total4)), dim3(NT), 0, (hipStream_t)stream, a, b, g, da, F, total4);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_sqdist_sum_fwd(const float* a, const float* b, int64_t n, float* term, float term_scale, float* total, float total_scale, void* stream) {
    if (!a || !b || n < 4 || (n & 3) || !aligned16(a) || !aligned16(b) || (!term && !total)) return EG3D_ERR_INVALID;
    const int bx = (int)std::min<int64_t>(512, (n / 4 + NT * 4 - 1) / (NT * 4));
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, term, 1); EG3D_DET_BIND(det, total, 1); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(sqdist_sum_fwd_kernel, dim3(std::max(bx, 1)), dim3(NT), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                       reinterpret_cast<const float4*>(b), n / 4, term, term_scale, total, total_scale);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_sqdist_sum_bwd(const float* a, const float* b, const float* g, float gscale, float* da, int64_t n, void* stream) {
    if (!a || !b || !g || !da || n < 4 || (n & 3) || !aligned16(a) || !aligned16(b) || !aligned16(da)) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(sqdist_sum_bwd_kernel, dim3(grid_blocks(n / 4)), dim3(NT), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                       reinterpret_cast<const float4*>(b), g, gscale, reinterpret_cast<float4*>(da), n / 4);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_tv_norm_fwd(const float* v, int B, int H, int W, float* term, float term_scale, float* total, float total_scale, void* stream) {
    if (!v || B < 1 || H < 2 || W < 2 || (!term && !total)) return EG3D_ERR_INVALID;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, term, 1); EG3D_DET_BIND(det, total, 1); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(tv_norm_fwd_kernel, dim3(grid_blocks((int64_t)B * H * W)), dim3(NT), 0, (hipStream_t)stream, v, B, H, W, term, term_scale, total,
                       total_scale);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_tv_norm_bwd(const float* v, const float* g, float gscale, float* dv, int B, int H, int W, void* stream) {
    if (!v || !g || !dv || B < 1 || H < 2 || W < 2) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(tv_norm_bwd_kernel, dim3(grid_blocks((int64_t)B * H * W)), dim3(NT), 0, (hipStream_t)stream, v, g, gscale, dv, B, H, W);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_slice_rgb4_fwd(const float* x, float* y4, int64_t P, int C, void* stream) {
    if (!x || !y4 || P < 1 || C < 4 || (C & 3) || !aligned16(x) || !aligned16(y4)) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(slice_rgb4_fwd_kernel, dim3((unsigned)((P + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, x, reinterpret_cast<float4*>(y4), P, C);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

static int slice_rgb4_bwd_impl(const float* dy4, const float* addend, float* dx, int64_t P, int C, void* stream) {
    if (!dy4 || !dx || P < 1 || C < 4 || (C & 3) || !aligned16(dx) || !aligned16(dy4) || !aligned16(addend)) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(slice_rgb4_bwd_kernel, dim3((unsigned)((P * (C / 4) + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(dy4), reinterpret_cast<const float4*>(addend), reinterpret_cast<float4*>(dx), P, C / 4);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_slice_rgb4_bwd(const float* dy4, float* dx, int64_t P, int C, void* stream) { return slice_rgb4_bwd_impl(dy4, nullptr, dx, P, C, stream); }

extern "C" int eg3d_slice_rgb4_bwd_add(const float* dy4, const float* addend, float* dx, int64_t P, int C, void* stream) {
    if (!addend) return EG3D_ERR_INVALID;
    return slice_rgb4_bwd_impl(dy4, addend, dx, P, C, stream);
}

extern "C" int eg3d_warp_project_fwd(const float* origins, const float* dirs, const float* depth, const float* consts, float* uv, int64_t P, void* stream) {
    if (!origins || !dirs || !depth || !consts || !uv || P < 1) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(warp_project_fwd_kernel, dim3((unsigned)((P + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, origins, dirs, depth, consts,
                       reinterpret_cast<float2*>(uv), P);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_warp_project_bwd(const float* origins, const float* dirs, const float* depth, const float* consts, const float* duv, float* d_origins,
                                     float* d_dirs, float* d_depth, int64_t P, void* stream) {
    if (!origins || !dirs || !depth || !consts || !duv || !d_origins || !d_dirs || !d_depth || P < 1) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(warp_project_bwd_kernel, dim3((unsigned)((P + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, origins, dirs, depth, consts,
                       reinterpret_cast<const float2*>(duv), d_origins, d_dirs, d_depth, P);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_grid_sample_nhwc_fwd(const float* input, const float* grid, float* out, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
    if (!input || !grid || !out || N < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || Ho < 1 || Wo < 1) return EG3D_ERR_INVALID;
    if (!aligned16(input) || !aligned16(out) || ((uintptr_t)grid & 7)) return EG3D_ERR_INVALID;
    const int64_t P = (int64_t)N * Ho * Wo;
    hipLaunchKernelGGL(grid_sample_nhwc_fwd_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, input, reinterpret_cast<const float2*>(grid), out,
                       P, (int64_t)Ho * Wo, H, W, C);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_grid_sample_nhwc_bwd(const float* input, const float* grid, const float* dout, float* dgrid, float* dinput, int N, int H, int W, int C, int Ho,
                                         int Wo, void* stream) {
    if (!input || !grid || !dout || !dgrid || N < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || Ho < 1 || Wo < 1) return EG3D_ERR_INVALID;
    if (!aligned16(input) || !aligned16(dout) || ((uintptr_t)grid & 7) || ((uintptr_t)dgrid & 7)) return EG3D_ERR_INVALID;
    const int64_t P = (int64_t)N * Ho * Wo;
    EG3D_DET_SCOPE(det, stream); EG3D_DET_BIND(det, dinput, (int64_t)N * H * W * C); EG3D_DET_COMMIT(det);
    hipLaunchKernelGGL(grid_sample_nhwc_bwd_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, input, reinterpret_cast<const float2*>(grid), dout,
                       reinterpret_cast<float2*>(dgrid), dinput, P, (int64_t)Ho * Wo, H, W, C);
    EG3D_DET_END(det);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
