// Pose chain of the latent projector as ONE launch per direction: pose vector (quaternion | 6-D | two Euler angles) + optimisable translation ->
// camera extrinsic [4,4] and conditioning vector c [25] (training/projectors/w_projector.py:147-172, utils/camera_utils.py:201-228,259-273,
// 241-257,158-188).  In PyTorch this is ~150 one-element kernels forward and as many backward (config C3: a third of the step's launches).
// One thread per batch element evaluates the chain in FORWARD-MODE dual numbers (value + the partial derivatives with respect to the <= 9
// inputs), so the launch returns the 12 non-constant outputs (rotation 3x3, translation 3) AND their Jacobian [12][9]; the backward launch is
// the transposed Jacobian-vector product with the incoming d ext + d c.  Same formulas, same order of operations as the reference functions.
#include "common.h"

namespace {

constexpr int NI = 9;          // inputs: pose (<= 6) then translation (3)

struct Dual {
    float v;
    float d[NI];
};
__device__ __forceinline__ Dual cst(float c) { Dual r; r.v = c; for (int i = 0; i < NI; ++i) r.d[i] = 0.f; return r; }
__device__ __forceinline__ Dual var(float x, int i) { Dual r = cst(x); r.d[i] = 1.f; return r; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < NI; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < NI; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < NI; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < NI; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ Dual operator*(float s, const Dual& a) { Dual r; r.v = s * a.v; for (int i = 0; i < NI; ++i) r.d[i] = s * a.d[i]; return r; }
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
    Dual r; r.v = a.v / b.v;
    const float inv = 1.f / b.v;
    for (int i = 0; i < NI; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ Dual dsqrt(const Dual& a) { Dual r; r.v = sqrtf(a.v); const float h = 0.5f / r.v; for (int i = 0; i < NI; ++i) r.d[i] = a.d[i] * h; return r; }
__device__ __forceinline__ Dual dsin(const Dual& a) { Dual r; r.v = sinf(a.v); const float c = cosf(a.v); for (int i = 0; i < NI; ++i) r.d[i] = a.d[i] * c; return r; }
__device__ __forceinline__ Dual dcos(const Dual& a) { Dual r; r.v = cosf(a.v); const float s = -sinf(a.v); for (int i = 0; i < NI; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ Dual clamp_min(const Dual& a, float m) { return a.v < m ? cst(m) : a; }          // torch.clamp(min=): zero gradient below

struct V3 { Dual x, y, z; };
__device__ __forceinline__ Dual dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 scale(const V3& a, const Dual& s) { return {a.x * s, a.y * s, a.z * s}; }
// F.normalize(v, eps): v / max(||v||, eps)
__device__ __forceinline__ V3 normalize(const V3& a, float eps) {
    const Dual n = clamp_min(dsqrt(dot(a, a)), eps);
    return {a.x / n, a.y / n, a.z / n};
}

// mode 0 quaternion (w, x, y, z), 1 six-dimensional, 2 Euler (theta, phi offsets from pi / 2)
__global__ void pose_chain_kernel(const float* __restrict__ pose, const float* __restrict__ trans, int B, int mode, float radius, float* __restrict__ out12,
                                  float* __restrict__ jac) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int np = mode == 0 ? 4 : (mode == 1 ? 6 : 2);
    Dual R[3][3];
    if (mode == 0) {
        Dual q[4];
        for (int i = 0; i < 4; ++i) q[i] = var(pose[b * 4 + i], i);
        const Dual n = dsqrt(clamp_min(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], 1e-8f));
        const Dual w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
        const Dual one = cst(1.f);
        R[0][0] = one - 2.f * (y * y) - 2.f * (z * z); R[0][1] = 2.f * (x * y) - 2.f * (z * w); R[0][2] = 2.f * (x * z) + 2.f * (y * w);
        R[1][0] = 2.f * (x * y) + 2.f * (z * w); R[1][1] = one - 2.f * (x * x) - 2.f * (z * z); R[1][2] = 2.f * (y * z) - 2.f * (x * w);
        R[2][0] = 2.f * (x * z) - 2.f * (y * w); R[2][1] = 2.f * (y * z) + 2.f * (x * w); R[2][2] = one - 2.f * (x * x) - 2.f * (y * y);
    } else if (mode == 1) {
        V3 v0 = {var(pose[b * 6 + 0], 0) + cst(1e-4f), var(pose[b * 6 + 1], 1) + cst(1e-4f), var(pose[b * 6 + 2], 2) + cst(1e-4f)};
        V3 v1 = {var(pose[b * 6 + 3], 3) + cst(1e-4f), var(pose[b * 6 + 4], 4) + cst(1e-4f), var(pose[b * 6 + 5], 5) + cst(1e-4f)};
        const V3 e1 = normalize(v0, 1e-12f);
        const Dual pr = dot(e1, v1);
        const V3 u = {v1.x - pr * e1.x, v1.y - pr * e1.y, v1.z - pr * e1.z};
        const V3 e2 = normalize(u, 1e-12f);
        const V3 e3 = cross(e1, e2);
        R[0][0] = e1.x; R[1][0] = e1.y; R[2][0] = e1.z;          // columns e1, e2, e1 x e2
        R[0][1] = e2.x; R[1][1] = e2.y; R[2][1] = e2.z;
        R[0][2] = e3.x; R[1][2] = e3.y; R[2][2] = e3.z;
    } else {
        const float hp = 1.5707963267948966f, pi = 3.141592653589793f;
        const Dual theta = cst(hp) + var(pose[b * 2 + 0], 0), phi = cst(hp) + var(pose[b * 2 + 1], 1);
        const Dual sp = dsin(phi);
        const Dual pt = cst(pi) - theta;
        const V3 origin = {radius * (sp * dcos(pt)), radius * dcos(phi), radius * (sp * dsin(pt))};
        const V3 fwd = normalize({-origin.x, -origin.y, -origin.z}, 0.f);
        const V3 upv = {cst(0.f), cst(1.f), cst(0.f)};
        const V3 rn = normalize(cross(upv, fwd), 0.f);
        const V3 right = {-rn.x, -rn.y, -rn.z};
        const V3 up = normalize(cross(fwd, right), 0.f);
        R[0][0] = right.x; R[1][0] = right.y; R[2][0] = right.z;
        R[0][1] = up.x; R[1][1] = up.y; R[2][1] = up.z;
        R[0][2] = fwd.x; R[1][2] = fwd.y; R[2][2] = fwd.z;
    }
    // pose_to_cam: t = normalise(-R trans * radius - radius R[:, 2]) * radius          (w_projector.py:160-172)
    Dual tr[3];
    for (int i = 0; i < 3; ++i) tr[i] = var(trans[b * 3 + i], np + i);
    V3 t;
    Dual* tp[3] = {&t.x, &t.y, &t.z};
    for (int r = 0; r < 3; ++r) {
        const Dual pred = (-radius) * R[r][2];
        const Dual tw = (-(R[r][0] * tr[0] + R[r][1] * tr[1] + R[r][2] * tr[2])) * cst(radius);
        *tp[r] = tw + pred;
    }
    const Dual nrm = dsqrt(dot(t, t));
    const V3 tt = {t.x / nrm * cst(radius), t.y / nrm * cst(radius), t.z / nrm * cst(radius)};
    const Dual* outs[12] = {&R[0][0], &R[0][1], &R[0][2], &R[1][0], &R[1][1], &R[1][2], &R[2][0], &R[2][1], &R[2][2], &tt.x, &tt.y, &tt.z};
    for (int o = 0; o < 12; ++o) {
        out12[b * 12 + o] = outs[o]->v;
        for (int i = 0; i < NI; ++i) jac[(b * 12 + o) * NI + i] = outs[o]->d[i];
    }
}

// d_in[b][j] = sum_o g12[b][o] * jac[b][o][j];  g12 = gradient with respect to (R row-major, t) gathered from d ext [B,16] and d c [B,25]
__global__ void pose_chain_bwd_kernel(const float* __restrict__ jac, const float* __restrict__ d_ext, const float* __restrict__ d_cam, int B, int np,
                                      float* __restrict__ d_pose, float* __restrict__ d_trans) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float g[12];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            float v = 0.f;
            if (d_ext) v += d_ext[b * 16 + r * 4 + c];
            if (d_cam) v += d_cam[b * 25 + r * 4 + c];
            g[c < 3 ? r * 3 + c : 9 + r] = v;
        }
    for (int j = 0; j < np + 3; ++j) {
        float s = 0.f;
        for (int o = 0; o < 12; ++o) s = fmaf(g[o], jac[(b * 12 + o) * NI + j], s);
        if (j < np) { if (d_pose) d_pose[b * np + j] = s; }
        else if (d_trans) d_trans[b * 3 + (j - np)] = s;
    }
}

// ext [B,16] and c [B,25] from the 12 outputs + the intrinsics
__global__ void pose_chain_assemble_kernel(const float* __restrict__ out12, const float* __restrict__ K9, int B, float* __restrict__ ext, float* __restrict__ cam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 25) return;
    const int b = i / 25, k = i - b * 25;
    float v;
    if (k < 16) {
        const int r = k >> 2, c = k & 3;
        v = r < 3 ? (c < 3 ? out12[b * 12 + r * 3 + c] : out12[b * 12 + 9 + r]) : (c == 3 ? 1.f : 0.f);
        ext[b * 16 + k] = v;
    } else {
        v = K9[k - 16];
    }
    cam[i] = v;
}

}  // namespace

extern "C" int eg3d_pose_chain_fwd(const float* pose, const float* translation, const float* intrinsics, int B, int mode, float radius, float* ext, float* cam,
                                   float* jac, float* out12, void* stream) {
    if (!pose || !translation || !intrinsics || !ext || !cam || !jac || !out12 || B <= 0 || mode < 0 || mode > 2) return EG3D_ERR_INVALID;
    hipLaunchKernelGGL(pose_chain_kernel, dim3(eg3d_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, pose, translation, B, mode, radius, out12, jac);
    hipLaunchKernelGGL(pose_chain_assemble_kernel, dim3(eg3d_cdiv(B * 25, 64)), dim3(64), 0, (hipStream_t)stream, out12, intrinsics, B, ext, cam);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}

extern "C" int eg3d_pose_chain_bwd(const float* jac, const float* d_ext, const float* d_cam, int B, int mode, float* d_pose, float* d_translation, void* stream) {
    if (!jac || B <= 0 || mode < 0 || mode > 2 || (!d_ext && !d_cam)) return EG3D_ERR_INVALID;
    const int np = mode == 0 ? 4 : (mode == 1 ? 6 : 2);
    hipLaunchKernelGGL(pose_chain_bwd_kernel, dim3(eg3d_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, jac, d_ext, d_cam, B, np, d_pose, d_translation);
    EG3D_LAUNCH_CHECK();
    return EG3D_OK;
}
