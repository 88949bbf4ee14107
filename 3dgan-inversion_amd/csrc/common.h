// Shared helpers for the gfx950 kernels of libeg3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <math.h>
#include "../../include/eg3d_hip.h"

#define EG3D_LAUNCH_CHECK()                          \
    do {                                             \
        hipError_t _e = hipGetLastError();           \
        if (_e != hipSuccess) return (int)_e;        \
    } while (0)

static inline int eg3d_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#ifdef __cplusplus
#include <atomic>
#include <type_traits>
#include <utility>

// Zero-fill of a small accumulator as a KERNEL node.  hipMemsetAsync becomes a memset node when the step is captured into a HIP graph, and
// the ROCm 7.0 runtime's packet-captured replay of a single-branch graph stops honouring such a node after the first device-wide
// synchronise (found with rocgdb: the tri-plane scatter's bin cursors were no longer cleared, the fill pass ran past its id buffer --
// "write access to a read-only page"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 also avoids it).  No entry point of this library issues memset nodes.
static __global__ void eg3d_zero_words_kernel(uint32_t* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static inline void eg3d_zero_words(void* p, int64_t nwords, hipStream_t st) {
    if (nwords > 0) hipLaunchKernelGGL(eg3d_zero_words_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), nwords);
}
// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute is a property of the (function, device) pair, so the "already
// done" memo is one bit per device ordinal: correct with several devices in one process and from several host threads (a lost race
// only repeats the idempotent call).  Returns 0 or a HIP error code.
static inline int eg3d_ensure_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const uint64_t bit = 1ull << (dev & 63);
    if (dev < 64 && (done_mask.load(std::memory_order_acquire) & bit)) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev < 64) done_mask.fetch_or(bit, std::memory_order_release);
    return 0;
}
#endif

// ---- activations (semantics of bias_act: forward, first and second derivative keyed on the output) ----------
__device__ __forceinline__ float eg3d_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float eg3d_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

#define EG3D_SELU_S 1.0507009873554804934193349852946f
#define EG3D_SELU_A 1.6732632423543772848170429916717f

__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ double m_expm1(double x) { return expm1(x); }
__device__ __forceinline__ float m_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double m_log1p(double x) { return log1p(x); }
__device__ __forceinline__ float m_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double m_tanh(double x) { return tanh(x); }

template <typename F>
__device__ __forceinline__ F eg3d_act_fwd(F x, int act, F alpha) {
    switch (act) {
        default:
        case EG3D_ACT_LINEAR: return x;
        case EG3D_ACT_RELU: return x > 0 ? x : (F)0;
        case EG3D_ACT_LRELU: return x > 0 ? x : x * alpha;
        case EG3D_ACT_TANH: return m_tanh(x);
        case EG3D_ACT_SIGMOID: return (F)1 / ((F)1 + m_exp(-x));
        case EG3D_ACT_ELU: return x >= 0 ? x : m_expm1(x);
        case EG3D_ACT_SELU: return x >= 0 ? x * (F)EG3D_SELU_S : (F)(EG3D_SELU_S * EG3D_SELU_A) * m_expm1(x);
        case EG3D_ACT_SOFTPLUS: return x > 20 ? x : m_log1p(m_exp(x));
        case EG3D_ACT_SWISH: return x / ((F)1 + m_exp(-x));
    }
}

// The three piecewise-linear activations (linear, relu, lrelu -- the only ones on the generator path) as one branch-free form:
// y = x > 0 ? x : slope * x with slope 1 / 0 / alpha; derivative (keyed on the output) y > 0 ? 1 : slope.  Kernels whose fused epilogue
// only ever sees these use it instead of the nine-way switch (which, inlined per element, made e.g. the FIR epilogue 7000 instructions).
__host__ __device__ __forceinline__ bool eg3d_act_is_pwl(int act) { return act == EG3D_ACT_LINEAR || act == EG3D_ACT_RELU || act == EG3D_ACT_LRELU; }
__host__ __device__ __forceinline__ float eg3d_act_pwl_slope(int act, float alpha) { return act == EG3D_ACT_LINEAR ? 1.f : (act == EG3D_ACT_LRELU ? alpha : 0.f); }
__device__ __forceinline__ float eg3d_pwl_fwd(float x, float slope) { return x > 0.f ? x : (slope == 0.f ? 0.f : x * slope); }
__device__ __forceinline__ float eg3d_pwl_d1(float yy, float slope) { return yy > 0.f ? 1.f : slope; }

// first derivative of act, expressed with the un-gained output yy = y/gain (x only for swish)
template <typename F>
__device__ __forceinline__ F eg3d_act_d1(F yy, F x, int act, F alpha) {
    switch (act) {
        default:
        case EG3D_ACT_LINEAR: return (F)1;
        case EG3D_ACT_RELU: return yy > 0 ? (F)1 : (F)0;
        case EG3D_ACT_LRELU: return yy > 0 ? (F)1 : alpha;
        case EG3D_ACT_TANH: return (F)1 - yy * yy;
        case EG3D_ACT_SIGMOID: return yy * ((F)1 - yy);
        case EG3D_ACT_ELU: return yy >= 0 ? (F)1 : yy + (F)1;
        case EG3D_ACT_SELU: return yy >= 0 ? (F)EG3D_SELU_S : yy + (F)(EG3D_SELU_S * EG3D_SELU_A);
        case EG3D_ACT_SOFTPLUS: return (F)1 - m_exp(-yy);
        case EG3D_ACT_SWISH: {
            if (x > 40) return (F)1;
            if (x < -80) return (F)0;
            F e = m_exp(x);
            F d = e + (F)1;
            return e * (x + d) / (d * d);
        }
    }
}

// second derivative of act
template <typename F>
__device__ __forceinline__ F eg3d_act_d2(F yy, F x, int act, F alpha) {
    switch (act) {
        default: return (F)0;
        case EG3D_ACT_TANH: return (F)-2 * yy * ((F)1 - yy * yy);
        case EG3D_ACT_SIGMOID: return yy * ((F)1 - yy) * ((F)1 - (F)2 * yy);
        case EG3D_ACT_ELU: return yy >= 0 ? (F)0 : yy + (F)1;
        case EG3D_ACT_SELU: return yy >= 0 ? (F)0 : yy + (F)(EG3D_SELU_S * EG3D_SELU_A);
        case EG3D_ACT_SOFTPLUS: {
            F c = m_exp(-yy);
            return c * ((F)1 - c);
        }
        case EG3D_ACT_SWISH: {
            if (x > 40 || x < -80) return (F)0;
            F e = m_exp(x);
            F d = e + (F)1;
            return e * ((F)2 * d + x * ((F)1 - e)) / (d * d * d);
        }
    }
}

// max|v| reporting: wave-level reduce, then ONE conditional atomic per wave -- thousands of unconditional atomics on one address
// serialise at ~12 ns each (measured: a 25 us pass became 77 us), so a wave first looks at the current value (monotone: a stale read
// only costs a redundant atomic) and almost every wave skips.  Non-negative floats order like their bit patterns.
__device__ __forceinline__ void eg3d_commit_amax(float m, float* out) {
    if (out == nullptr) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f && m < 3.0e38f) {
        const float cur = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m > cur) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
    }
}

// Block-level form (every thread of the block must call it): wave maxima meet in LDS, ONE conditional atomic per block.  For short
// memory-bound kernels all waves finish within the same few microseconds, every one reads the still-stale value and the per-wave form
// degenerates into thousands of serialised atomics on one address (a 20 us pass took 55 us).
__device__ __forceinline__ void eg3d_commit_amax_block(float m, float* out, const int nwaves = 0 /* waves taking part (0 = blockDim.x / 64) */) {
    if (out == nullptr) return;                 // uniform across the block
    __shared__ float s_wave_max[16];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int nw = nwaves > 0 ? nwaves : (int)((blockDim.x + 63) >> 6);
    if ((threadIdx.x & 63) == 0) s_wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, s_wave_max[w]);
        if (m > 0.f && m < 3.0e38f) {
            const float cur = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (m > cur) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
        }
    }
}

// ---- EG3D_EPI_BWD_ACT: the producing layer's activation backward inside a data-gradient epilogue (conv_igemm.hip, conv_v2.hip) --------
// One float4 unit (4 channels of one pixel): v = that layer's dout, o = its saved output.  Returns dz = dy * d; accumulates the
// per-channel sums (accb: dy, accd: dy * (pre - bias - noise)) and hands back the unit's channel sum of dy.
struct eg3d_act_bwd_consts {
    float slope, gain, inv_gain, clamp, strength, inv_slope;
};
__device__ __forceinline__ eg3d_act_bwd_consts eg3d_act_bwd_setup(const eg3d_act_bwd& ab) {
    eg3d_act_bwd_consts c;
    c.slope = eg3d_act_pwl_slope(ab.act, ab.alpha);
    c.gain = ab.gain; c.inv_gain = 1.f / ab.gain; c.clamp = ab.clamp;
    c.inv_slope = c.slope != 0.f ? 1.f / c.slope : 0.f;           // (a product instead of an IEEE division per element: <= 1 ulp, in `pre` only)
    c.strength = (ab.noise != nullptr && ab.noise_strength != nullptr) ? *ab.noise_strength : 0.f;
    return c;
}
__device__ __forceinline__ float4 eg3d_act_bwd_unit(const eg3d_act_bwd_consts& c, float4 v, float4 o, float4 d4, float4 b4, float nzs,
                                                    float4& accb, float4& accd, float& chan_sum) {
    const float vv[4] = {v.x, v.y, v.z, v.w}, oo[4] = {o.x, o.y, o.z, o.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
    float dy[4], ad[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float yy = oo[q] * c.inv_gain;
        float g = vv[q] * c.gain * (yy > 0.f ? 1.f : c.slope);
        if (c.clamp >= 0.f && (oo[q] >= c.clamp || oo[q] <= -c.clamp)) g = 0.f;
        const float pre = (yy > 0.f || c.slope == 0.f) ? yy : yy * c.inv_slope;
        dy[q] = g;
        ad[q] = g * (pre - bb[q] - nzs);
    }
    accb.x += dy[0]; accb.y += dy[1]; accb.z += dy[2]; accb.w += dy[3];
    accd.x += ad[0]; accd.y += ad[1]; accd.z += ad[2]; accd.w += ad[3];
    chan_sum = (dy[0] + dy[1]) + (dy[2] + dy[3]);
    return make_float4(dy[0] * d4.x, dy[1] * d4.y, dy[2] * d4.z, dy[3] * d4.w);
}
// sum over the `group` (power of two <= 64) consecutive lanes that hold the channel quads of one pixel row
// (valid in every lane of the group).  Within a row of 16 lanes the exchange is four DPP modifiers on the adds -- quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror -- instead of four ds_bpermute round trips through the LDS crossbar; only the 16 / 32 strides still shuffle.
template <int CTRL>
__device__ __forceinline__ float eg3d_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float eg3d_row_group_sum(float s, int group) {
    if (group >= 2) s += eg3d_dpp<0xB1>(s);          // quad_perm [1,0,3,2]
    if (group >= 4) s += eg3d_dpp<0x4E>(s);          // quad_perm [2,3,0,1]
    if (group >= 8) s += eg3d_dpp<0x141>(s);         // row_half_mirror: lane i <-> 7 - i
    if (group >= 16) s += eg3d_dpp<0x140>(s);        // row_mirror: lane i <-> 15 - i
    if (group >= 32) s += __shfl_xor(s, 16);
    if (group >= 64) s += __shfl_xor(s, 32);
    return s;
}

// XCD-aware block-id remap (bijective for any grid size): hardware places block b on XCD b % 8; give each XCD a
// contiguous chunk of logical tiles so that tiles sharing operands share an L2.
__device__ __forceinline__ int eg3d_xcd_remap(int bid, int nblocks) {
    const int NX = 8;
    int q = nblocks / NX, r = nblocks % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- the step boundary of every LDS-DMA pipeline of this family (conv_v2 / conv_v2_s2adj / conv_v2_up / conv_wgrad_v2) -------------------------------------
// A step reads LDS tiles that LDS-DMA operations of ALL waves of the workgroup brought in, and issues LDS-DMA operations into slots that were READ during the
// previous step.  One boundary per step makes both directions safe:
//     s_waitcnt vmcnt(N)     this wave's LDS-DMA operations, except the youngest N, have written LDS            (read-after-write: the tiles of this step)
//     s_waitcnt lgkmcnt(0)   this wave's LDS reads of the previous step have RETURNED                             (write-after-read: the slots re-used from here on)
//     s_barrier              ... and so have every other wave's
// in ONE asm statement: nothing can be scheduled between the waits and the barrier, and the "memory" clobber keeps every LDS access and every LDS-DMA issue on
// the side of the boundary the source puts it on.
//
// ROOT CAUSE of the two "schedule-dependent" faults of rounds 2 and 5 (tools/rootcause/): until round 6 the boundary was
//     asm volatile("s_waitcnt vmcnt(N)" ::: "memory");  __builtin_amdgcn_s_barrier();
// The builtin is IntrNoMem: it orders nothing but itself.  The scheduler sinks the last matrix instructions of a step -- and the `s_waitcnt lgkmcnt` in front of
// them -- BELOW the barrier, so 2 .. 7 ds_read_b128 of the previous step (in the 4- and 2-row instantiations the low pieces and the second high piece of the weight
// ring) were still IN FLIGHT when a wave arrived at the barrier.  Another wave then passes the barrier and issues the LDS-DMA of step + 2 into the ring slot
// those reads address; weight tiles are L2 / TCP hits, the DMA can land before a read that is queued behind the other workgroup's LDS traffic: the read
// returns the tile of the wrong tap.  Low pieces are read last, hence errors of relative size 2^-11 / steps = 1e-7 .. 1e-5, sporadic, and dependent on
// exactly how many reads the scheduler happened to leave behind the barrier -- which any neutral change of the source re-rolls.
// tools/rootcause/isa_protocol.py checks the compiled code of every kernel that uses this boundary (no LDS read in flight at a barrier, LDS-DMA issues per
// step and vmcnt immediates equal to the constexpr schedule below); tests/test_isa_protocol.py runs it on every build.
template <int N>
__device__ __forceinline__ void step_sync() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate (6 bits on gfx9)");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

