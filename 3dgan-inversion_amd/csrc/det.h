// Order-independent accumulation: the second build of the library (make det -> inv3d_amd/libeg3d_hip_det.so, -DEG3D_DET=1).
//
// Every floating-point atomic of the library goes through eg3d_acc().  In the normal build that IS unsafeAtomicAdd (same code as
// before this header existed).  In the deterministic build a value is added, as an exact integer, to a small fixed-point
// super-accumulator (four 64-bit words per target: bits 2^-90 .. 2^70 of the sum in 40-bit digits with 24 bits of carry room each), so
// the result does not depend on the order in which blocks, waves or lanes arrive -- integer addition is associative -- and is the
// correctly rounded sum up to one final double -> float rounding, PROVIDED no digit receives more than 2^23 same-sign additions between two
// closes of a scope (24 bits of carry room; the largest target of the library, a 512^2 plane-gradient texel row, sees < 2^21) and every
// |value| < 2^53 (larger ones fall back to the float atomic and count in eg3d_det_misses()).  That is stronger than an ordered reduction: the same bits come out
// of a different grid, a different split-K factor or a different sort order of the renderer's scatter lists.
//
// Who owns the accumulators: the library call.  An entry point that accumulates opens a scope, binds its targets (pointer + element
// count; any memory -- the caller's own tensors), launches its kernels, and closes the scope: the closing kernel converts each
// accumulator to float, ADDS it to the target (targets may hold a starting value) and clears it.  The words live in a workspace the
// caller lends once (eg3d_det_set_workspace); a call's targets must fit in it (EG3D_ERR_WORKSPACE otherwise).  A target that was not
// bound falls back to the float atomic and counts in eg3d_det_misses() -- the tests assert that count stays zero.
//
// One stream at a time: the bound-region table is a single device object updated in stream order.
//
// The reference has no counterpart (its backward passes are PyTorch's: cuDNN / atomicAdd, non-deterministic the same way); the mode
// exists so that the parity tests can hold bounds that are not widened by run-to-run noise.
#pragma once
#include "common.h"

#ifndef EG3D_DET
#define EG3D_DET 0
#endif

#if !EG3D_DET

__device__ __forceinline__ void eg3d_acc(float* p, float v) { unsafeAtomicAdd(p, v); }
// a block-local (LDS) partial sum that is later committed to `gp` by one thread: the normal build accumulates in LDS
#define EG3D_LDS_ACC(lds_ptr, gp, v) atomicAdd((lds_ptr), (v))
#define EG3D_DET_SCOPE(name, stream)
#define EG3D_DET_BIND(name, ptr, count)
#define EG3D_DET_COMMIT(name)
#define EG3D_DET_FLUSH(name)
#define EG3D_DET_END(name)
#define EG3D_DET_BIND_ACT(name, ab, N, C, HW)
#define EG3D_DET_DIV(v, d) (v)

#else

constexpr int EG3D_DET_NW = 4;           // 64-bit words per target
constexpr int EG3D_DET_DIGIT = 40;       // bits of the sum a word is the home of (the other 24: carries of up to 2^23 additions)
constexpr int EG3D_DET_OFF = 90;         // weight of bit 0: 2^-90
constexpr int EG3D_DET_MAXR = 40;        // targets one call can bind

struct eg3d_det_region { float* base; unsigned long long count; long long* words; };       // words[k * count + i], k < EG3D_DET_NW
struct eg3d_det_table { int n; unsigned misses; eg3d_det_region r[EG3D_DET_MAXR]; };

// this translation unit's pointer to the table (set by eg3d_det_set_workspace through the registration below)
static __device__ eg3d_det_table* g_eg3d_det_tab = nullptr;

extern "C" void eg3d_det_register_tu(int (*set)(eg3d_det_table*));
namespace {
struct eg3d_det_tu_registration {
    eg3d_det_tu_registration() {
        eg3d_det_register_tu(+[](eg3d_det_table* t) -> int { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_eg3d_det_tab), &t, sizeof(t)); });
    }
} eg3d_det_tu_registration_instance;
}

__device__ __forceinline__ void eg3d_acc(float* p, float v) {
    eg3d_det_table* T = g_eg3d_det_tab;
    if (T == nullptr) { unsafeAtomicAdd(p, v); return; }
    const unsigned u = __float_as_uint(v);
    int e = (int)((u >> 23) & 0xffu);
    unsigned long long m = u & 0x7fffffu;
    if (e == 0) { if (m == 0) return; e = 1; } else m |= 0x800000u;             // v = +-m 2^(e - 150)
    const int n = T->n;
    long long* words = nullptr;
    unsigned long long cnt = 0, off = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long o = (unsigned long long)(p - T->r[i].base);
        if (o < T->r[i].count) { words = T->r[i].words; cnt = T->r[i].count; off = o; }
    }
    int b = e - 150 + EG3D_DET_OFF;                                              // bit index of the mantissa's lowest bit
    if (e == 255 || words == nullptr || b >= (EG3D_DET_NW - 1) * EG3D_DET_DIGIT) {
        unsafeAtomicAdd(p, v);                                                  // Inf / NaN poison the target as they would have; |v| >= 2^53 or an unbound target: counted
        if (e != 255) atomicAdd(&T->misses, 1u);
        return;
    }
    if (b < 0) {                                                                // below 2^-90: truncated toward zero (a function of the value alone)
        if (b <= -24) return;
        m >>= -b; b = 0;
        if (m == 0) return;
    }
    const int k = b / EG3D_DET_DIGIT, s = b - k * EG3D_DET_DIGIT;
    const unsigned long long x = m << s;                                        // < 2^63
    long long lo = (long long)(x & ((1ull << EG3D_DET_DIGIT) - 1)), hi = (long long)(x >> EG3D_DET_DIGIT);
    if (u >> 31) { lo = -lo; hi = -hi; }
    if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(words + (unsigned long long)k * cnt + off), (unsigned long long)lo);
    if (hi) atomicAdd(reinterpret_cast<unsigned long long*>(words + (unsigned long long)(k + 1) * cnt + off), (unsigned long long)hi);
}
#define EG3D_LDS_ACC(lds_ptr, gp, v) eg3d_acc((gp), (v))
// a value the normal build divides once, after the block's partial sums met in LDS: here every contribution is divided
#define EG3D_DET_DIV(v, d) ((v) / (d))

struct eg3d_det_host { eg3d_det_table* table; long long* words; unsigned long long nwords; int depth; };
extern "C" eg3d_det_host* eg3d_det_host_state();
extern "C" void eg3d_det_launch_bind(const eg3d_det_table* t, void* stream);
extern "C" void eg3d_det_launch_finalize(const eg3d_det_table* t, void* stream);

// Host side of one call.  Nested calls (an entry point calling another one) bind nothing: the outermost scope owns the table.
struct eg3d_det_scope_t {
    void* st;
    eg3d_det_table t;
    bool on, bad, outer;
    explicit eg3d_det_scope_t(void* stream) : st(stream), bad(false) {
        t.n = 0; t.misses = 0;
        eg3d_det_host* h = eg3d_det_host_state();
        outer = h->depth++ == 0;
        on = outer && h->table != nullptr;
    }
    ~eg3d_det_scope_t() { end(); eg3d_det_host_state()->depth--; }         // (an early error return still closes the scope)
    void bind(const float* p, long long count) {
        if (!on || p == nullptr || count <= 0) return;
        if (t.n == EG3D_DET_MAXR) { bad = true; return; }
        t.r[t.n].base = const_cast<float*>(p); t.r[t.n].count = (unsigned long long)count; t.r[t.n].words = nullptr;
        ++t.n;
    }
    int commit() {
        if (!on) return EG3D_OK;
        if (bad) return EG3D_ERR_WORKSPACE;
        if (!t.n) return EG3D_OK;
        // the same pointer bound twice, overlapping or exactly adjacent targets become one region.  (Gaps are NOT bridged: a stray eg3d_acc to
        // memory between two separate targets must count in `misses`, not be absorbed.)
        for (int i = 1; i < t.n; ++i)
            for (int j = i; j > 0 && t.r[j].base < t.r[j - 1].base; --j) { const eg3d_det_region x = t.r[j]; t.r[j] = t.r[j - 1]; t.r[j - 1] = x; }
        int m = 0;
        for (int i = 1; i < t.n; ++i) {
            float* end = t.r[m].base + t.r[m].count;
            if (t.r[i].base <= end) {
                float* e2 = t.r[i].base + t.r[i].count;
                if (e2 > end) t.r[m].count = (unsigned long long)(e2 - t.r[m].base);
            } else {
                t.r[++m] = t.r[i];
            }
        }
        t.n = m + 1;
        eg3d_det_host* h = eg3d_det_host_state();
        unsigned long long used = 0;
        for (int i = 0; i < t.n; ++i) {
            const unsigned long long need = (unsigned long long)EG3D_DET_NW * t.r[i].count;
            if (used + need > h->nwords) { bad = true; t.n = 0; return EG3D_ERR_WORKSPACE; }
            t.r[i].words = h->words + used;
            used += need;
        }
        eg3d_det_launch_bind(&t, st);
        return EG3D_OK;
    }
    void flush() { if (on && !bad && t.n) eg3d_det_launch_finalize(&t, st); }
    void end() {
        if (!on || bad || !t.n) return;
        eg3d_det_launch_finalize(&t, st);
        eg3d_det_table none; none.n = 0; none.misses = 0;
        eg3d_det_launch_bind(&none, st);
        t.n = 0;
    }
};
#define EG3D_DET_SCOPE(name, stream) eg3d_det_scope_t name(stream)
#define EG3D_DET_BIND(name, ptr, count) name.bind((ptr), (long long)(count))
#define EG3D_DET_COMMIT(name) do { if (int eg3d_det_rc = name.commit()) return eg3d_det_rc; } while (0)
#define EG3D_DET_FLUSH(name) name.flush()
#define EG3D_DET_END(name) name.end()
// the four reduction targets of a fused activation backward (eg3d_act_bwd): [C], [N,C], [*,HW] with batch stride dnoise_nstride, scalar
#define EG3D_DET_BIND_ACT(name, ab, N, C, HW) do { name.bind((ab).dbias, (long long)(C)); name.bind((ab).dd, (long long)(N) * (C)); \
        name.bind((ab).dnoise, (long long)((N) - 1) * (ab).dnoise_nstride + (HW)); name.bind((ab).dstrength, 1); } while (0)

#endif
