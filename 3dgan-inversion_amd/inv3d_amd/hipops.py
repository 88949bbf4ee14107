"""Thin, allocation-aware wrappers over the C-ABI kernels (one Python function per entry point).

Tensors handed to the fused path are fp32 "NHWC": logical shape [N,C,H,W] with channels_last strides (so the
reference-facing API keeps NCHW shapes while memory is channel-contiguous, which is what the MFMA implicit GEMM,
the float4 epilogues and the 128-byte tri-plane gathers want).
"""
import contextlib
import gc
import math
import os
import weakref
import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L

CL = torch.channels_last


class LaunchProfiler:
    """Optional per-launch timing (bench.py roofline): HIP events recorded on the launch stream around every implicit-GEMM conv launch,
    with the launch's algorithmic FLOPs and kernel id (tile configuration 0..4 of eg3d_conv2d_igemm_f32, V2_CONFIG for the pre-split
    kernel), and around named spans (the renderer's forward / backward)."""

    def __init__(self, only_config=None, keep_meta=False):
        self.records = []          # ((config_id, precision_id), algo_flops, start_event, end_event): one key per kernel family
        self.spans = []            # (name, start_event, end_event)
        self.only_config = only_config      # time only launches of this kernel id (keeps the event overhead small)
        self.meta = [] if keep_meta else None      # per record: launch geometry (tools/conv_launch_table.py)

    def summary(self):
        out = {}
        for cfg, fl, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = out.setdefault(cfg, dict(launches=0, flops=0.0, ms=0.0))
            d['launches'] += 1
            d['flops'] += fl
            d['ms'] += ms
        return out

    def span_summary(self):
        out = {}
        for name, e0, e1 in self.spans:
            d = out.setdefault(name, dict(count=0, ms=0.0))
            d['count'] += 1
            d['ms'] += e0.elapsed_time(e1)
        return out


class _Span:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prof = PROFILER
        if self.prof is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.prof is not None:
            self.e1.record()
            self.prof.spans.append((self.name, self.e0, self.e1))


def span_between_grads(name, t_start, t_end):
    """Profiler span over a stretch of the BACKWARD pass: opened when the gradient of `t_start` arrives, closed when that of `t_end` does
    (bench.py: the backbone's backward = from d planes to d ws).  No-op without an active profiler."""
    prof = PROFILER
    if prof is None or not (torch.is_tensor(t_start) and torch.is_tensor(t_end) and t_start.requires_grad and t_end.requires_grad):
        return
    st = {}

    def opened(g):
        st['e0'] = torch.cuda.Event(enable_timing=True)
        st['e0'].record()

    def closed(g):
        if 'e0' in st:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            prof.spans.append((name, st.pop('e0'), e1))
    t_start.register_hook(opened)
    t_end.register_hook(closed)


TILE_NAMES = {0: '128,128,2,2', 1: '64,128,2,2', 2: '32,128,1,4', 3: '128,32,4,1', 4: '256,64,4,1'}


PROFILER = None


def is_cl(t: torch.Tensor) -> bool:
    return t.dim() == 4 and t.is_cuda and t.dtype == torch.float32 and (t.stride(1) == 1 or t.shape[1] == 1) and \
        t.is_contiguous(memory_format=CL)


@contextlib.contextmanager
def capture_guard():
    """Wrap a HIP-graph capture: the cyclic garbage collector must not run inside it.  Collecting the remains of an EARLIER capture (a
    CUDAGraph, tensors of its pool) frees device memory, which the runtime refuses while a stream is capturing -- the process aborts
    ("Fatal Python error: Aborted ... Garbage-collecting", seen once in four runs of the GPU suite).  Collect before, hold during."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def to_cl(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous(memory_format=CL)


def empty_cl(n, c, h, w, device) -> torch.Tensor:
    return torch.empty((n, c, h, w), dtype=torch.float32, device=device, memory_format=CL)


class ZeroArena:
    """One zero-fill launch per optimisation step instead of ~60.

    Split-K partial-sum buffers and the small atomically accumulated outputs (style / noise / bias gradients ...) all have to start
    at zero, and inside a replayed graph every fill is a ~5 us kernel.  The arena hands out slices of ONE buffer that is cleared once
    at the start of the step; the demand of the previous step sizes it (shapes are static from step to step; anything that does not
    fit falls back to torch.zeros and enlarges the arena for the next step).  Slices are only valid until the next begin(): use it
    for temporaries and for gradients that are consumed within the step (inversion.LatentProjector does)."""

    def __init__(self, device):
        self.device, self.buf, self.off, self.limit, self.demand, self.prev_demand = device, None, 0, 0, 0, 0

    def begin(self):
        need = self.prev_demand
        if need and (self.buf is None or self.buf.numel() < need):
            self.buf = torch.empty(need, dtype=torch.float32, device=self.device)
        self.limit = need if self.buf is not None else 0
        if self.limit:
            self.buf[:self.limit].fill_(0.0)         # a fill kernel, not a memset node: memset / memcpy nodes split a captured graph into
                                                     # separately submitted segments (~50 us of idle GPU each on ROCm 7)
        self.off = self.demand = 0

    def take(self, numel: int):
        n = -(-numel // 64) * 64                        # 256-byte granules
        self.demand += n
        if self.off + n > self.limit:
            return None
        v = self.buf[self.off:self.off + numel]
        self.off += n
        return v

    def end(self):
        self.prev_demand = max(self.prev_demand, self.demand)


ARENA: Optional[ZeroArena] = None


class zero_arena:
    """Context manager: route hipops.zeros / zeros_cl through `arena` for the duration of one step."""

    def __init__(self, arena: Optional[ZeroArena]):
        self.arena = arena

    def __enter__(self):
        global ARENA
        self.prev, ARENA = ARENA, self.arena
        if self.arena is not None:
            self.arena.begin()
        return self.arena

    def __exit__(self, *exc):
        global ARENA
        if self.arena is not None:
            self.arena.end()
        ARENA = self.prev
        return False


def zeros(shape, device) -> torch.Tensor:
    """fp32 zeros; from the step's ZeroArena when one is active."""
    shape = tuple(shape)
    numel = 1
    for d in shape:
        numel *= int(d)
    a = ARENA.take(numel) if (ARENA is not None and torch.device(device) == ARENA.device and numel > 0) else None
    return a.view(shape) if a is not None else torch.zeros(shape, dtype=torch.float32, device=device)


def zeros_cl(n, c, h, w, device) -> torch.Tensor:
    a = ARENA.take(n * c * h * w) if (ARENA is not None and torch.device(device) == ARENA.device) else None
    if a is not None:
        return a.view(n, h, w, c).permute(0, 3, 1, 2)           # NHWC memory, NCHW shape = channels_last
    return torch.empty((n, c, h, w), dtype=torch.float32, device=device, memory_format=CL).zero_()


def _dtype_code(t):
    try:
        return {torch.float32: L.F32, torch.float16: L.F16, torch.float64: L.F64}[t.dtype]
    except KeyError:
        raise L.Eg3dHipError(f'unsupported dtype {t.dtype}')


# ------------------------------------------------------------------------------------------------- bias_act
def bias_act_raw(x, b, xref, yref, dy, grad, dim, act_id, alpha, gain, clamp):
    """y = eg3d_bias_act(...); x dense (any layout); output has x's layout."""
    L.require_cuda(x, b, xref, yref, dy)
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    size_b = b.numel() if b is not None else 0
    step_b = x.stride(dim) if b is not None else 1
    L.check(L.lib().eg3d_bias_act(L.ptr(x), L.ptr(b), L.ptr(xref), L.ptr(yref), L.ptr(dy), L.ptr(y), _dtype_code(x), x.numel(),
                                  size_b, step_b, grad, act_id, float(alpha), float(gain), float(clamp), L.stream_ptr()), 'bias_act')
    return y


# ------------------------------------------------------------------------------------------------- upfirdn2d
def upfirdn2d_raw(x, f2d, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    L.require_cuda(x, f2d)
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    ow = (w * upx + padx0 + padx1 - fw + downx) // downx
    oh = (h * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise L.Eg3dHipError('upfirdn2d: output must be at least 1x1')
    mf = CL if (x.stride(1) == 1 and c > 1) else torch.contiguous_format
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device, memory_format=mf)
    xs = (C.c_int64 * 4)(*x.stride())
    ys = (C.c_int64 * 4)(*y.stride())
    f2d = f2d.contiguous().float()
    L.check(L.lib().eg3d_upfirdn2d(L.ptr(x), L.ptr(f2d), L.ptr(y), _dtype_code(x), n, c, h, w, xs, fh, fw, upx, upy, downx, downy,
                                   padx0, padx1, pady0, pady1, int(bool(flip)), float(gain), oh, ow, ys, L.stream_ptr()), 'upfirdn2d')
    return y


def upfirdn2d_nhwc(x, f2d, up=1, down=1, pad=(0, 0, 0, 0), flip=False, gain=1.0, out=None, accumulate=False, addend=None):
    """addend (fp32 channels_last, the output's shape): returns result + addend in the same pass (eg3d_upfirdn2d_nhwc_add)."""
    assert is_cl(x), 'upfirdn2d_nhwc expects an fp32 channels_last CUDA tensor'
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    px0, px1, py0, py1 = pad
    ow = (w * up + px0 + px1 - fw + down) // down
    oh = (h * up + py0 + py1 - fh + down) // down
    if out is None:
        out = empty_cl(n, c, oh, ow, x.device)
        accumulate = False
    if addend is not None:
        assert not accumulate and is_cl(addend) and tuple(addend.shape) == (n, c, oh, ow) and addend.dtype == torch.float32
        L.check(L.lib().eg3d_upfirdn2d_nhwc_add(L.ptr(x), L.ptr(f2d), L.ptr(addend), L.ptr(out), n, c, h, w, fh, fw, up, down, px0, px1, py0, py1,
                                                int(bool(flip)), float(gain), oh, ow, L.stream_ptr()), 'upfirdn2d_nhwc_add')
        return out
    L.check(L.lib().eg3d_upfirdn2d_nhwc(L.ptr(x), L.ptr(f2d), L.ptr(out), n, c, h, w, fh, fw, up, down, px0, px1, py0, py1,
                                        int(bool(flip)), float(gain), oh, ow, int(bool(accumulate)), L.stream_ptr()), 'upfirdn2d_nhwc')
    return out


# ------------------------------------------------------------------------------------------------- conv tap lists
def _mk_class(Ha, Wa, py, px, taps):
    c = L.ConvClass()
    c.Ha, c.Wa, c.out_py, c.out_px, c.ntaps = Ha, Wa, py, px, len(taps)
    assert 1 <= len(taps) <= 9
    for i, (dy, dx, wt) in enumerate(taps):
        c.dy[i], c.dx[i], c.wtap[i] = dy, dx, wt
    return c


def classes_corr(Ho, Wo, kh, kw, pad, flip_taps=False, dil=1):
    """stride-1 correlation: out[y,x] = sum_k w[ky,kx] * in[y+dil*ky-pad, x+dil*kx-pad]  (flip_taps: true convolution; dil: dilation)."""
    taps = []
    for ky in range(kh):
        for kx in range(kw):
            wt = ((kh - 1 - ky) * kw + (kw - 1 - kx)) if flip_taps else (ky * kw + kx)
            taps.append((dil * ky - pad, dil * kx - pad, wt))
    return [_mk_class(Ho, Wo, 0, 0, taps)]


def classes_corr_adjoint(Hi, Wi, kh, kw, pad, flip_taps=False, dil=1):
    """data-gradient of classes_corr: dx[y,x] = sum_k w[ky,kx] * g[y-dil*ky+pad, x-dil*kx+pad]."""
    taps = []
    for ky in range(kh):
        for kx in range(kw):
            wt = ((kh - 1 - ky) * kw + (kw - 1 - kx)) if flip_taps else (ky * kw + kx)
            taps.append((pad - dil * ky, pad - dil * kx, wt))
    return [_mk_class(Hi, Wi, 0, 0, taps)]


def classes_convT(Hi, Wi, kh, kw, up, flip_taps=False):
    """stride-`up` transposed conv, no padding: out[up*a+ky, up*b+kx] += in[a,b]*w[ky,kx]; out = (Hi-1)*up + kh.
    One class per output phase (py,px); taps ky == py (mod up) read in[a - (ky-py)/up]."""
    Ho, Wo = (Hi - 1) * up + kh, (Wi - 1) * up + kw
    cls = []
    for py in range(up):
        for px in range(up):
            taps = []
            for ky in range(py, kh, up):
                for kx in range(px, kw, up):
                    wt = ((kh - 1 - ky) * kw + (kw - 1 - kx)) if flip_taps else (ky * kw + kx)
                    taps.append((-(ky - py) // up, -(kx - px) // up, wt))
            Ha = (Ho - py + up - 1) // up
            Wa = (Wo - px + up - 1) // up
            if taps and Ha > 0 and Wa > 0:
                cls.append(_mk_class(Ha, Wa, py, px, taps))
    return cls, Ho, Wo


def classes_convT_adjoint(Hi, Wi, kh, kw, up, flip_taps=False):
    """data-gradient of classes_convT: dx[a,b] = sum_k g[up*a+ky, up*b+kx] * w[ky,kx]  (in_stride = up)."""
    taps = []
    for ky in range(kh):
        for kx in range(kw):
            wt = ((kh - 1 - ky) * kw + (kw - 1 - kx)) if flip_taps else (ky * kw + kx)
            taps.append((ky, kx, wt))
    return [_mk_class(Hi, Wi, 0, 0, taps)]


_MEMO = {}
WEIGHTS_EPOCH = 0


def weights_changed():
    """Invalidate every derived weight image (memo(), fused.WeightCache).  The caches key on a tensor's in-place version counter, which
    ordinary in-place updates bump -- but fused multi-tensor optimisers (torch.optim.Adam(fused=True)) write parameters without
    touching it.  Call this after such an update (inversion.PivotalTuner registers it as an optimizer step hook)."""
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1



CAPTURE_REFS = None           # a list while graphed._capture runs: every derived tensor handed out (memo / WeightCache) is appended, so that the
                              # captured graph's entry can keep alive what its kernels hold raw pointers to


def keep_for_capture(*objs):
    if CAPTURE_REFS is not None:
        CAPTURE_REFS.extend(o for o in objs if o is not None)


def memo(tag, tensors, fn):
    """Derived images of parameters (packed / padded / pre-scaled weights), recomputed only when a source tensor changes (storage
    pointer or in-place version).  One entry per (tag, storage): frozen weights cost nothing per step, trained ones are rebuilt."""
    # the optimiser epoch only concerns tensors an optimiser can write: frozen sources (the loss / pose / encoder networks during
    # pivotal tuning) keep their images across the tuned generator's steps
    epoch = WEIGHTS_EPOCH if any(t.requires_grad for t in tensors) else -1
    key = (epoch,) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)
    slot = (tag, tensors[0].data_ptr())
    hit = _MEMO.get(slot)
    # the entry is only valid for the very same tensor objects (a freed temporary's address can be reused by another tensor)
    if hit is None or hit[0] != key or any(r() is not t for r, t in zip(hit[2], tensors)):
        if len(_MEMO) > 512:
            _MEMO.clear()
        with torch.no_grad():
            hit = (key, fn(), tuple(weakref.ref(t) for t in tensors))
        _MEMO[slot] = hit
    keep_for_capture(hit[1])
    return hit[1]


def pack_weight_fwd(w):
    """[O,I,kh,kw] -> [O, kh*kw*I]  (row o: taps major, input channel minor)."""
    o, i, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(o, kh * kw * i).contiguous()


def pack_weight_adj(w):
    """[O,I,kh,kw] -> [I, kh*kw*O]  (for data gradients: roles of I and O swapped)."""
    o, i, kh, kw = w.shape
    return w.permute(1, 2, 3, 0).reshape(i, kh * kw * o).contiguous()


# Matrix-core arithmetic of the implicit GEMMs (include/eg3d_hip.h EG3D_PREC_*).  Operands, accumulators and results are fp32 in
# every mode; the modes differ in how an fp32 product is formed on the matrix cores:
#   'f16x3'   two fp16 pieces per operand, three products; error vs fp64 as small as the fp32 MFMA path for operands of ordinary
#             magnitude (|x| ~ 2^-4 .. 2^16), 1.45x the rate of bf16x6.  Needs the operand range: used where it is known.
#   'bf16x6'  three bf16 pieces, six products; fp32-equivalent for any operand range.
#   'f32'     v_mfma_f32_32x32x2_f32.       'bf16x3'  three bf16 products (~2^-15), opt-in only.
# Mode 'auto' (default): the style-modulated convolutions run 'f16x3' -- their forward operands are demodulated O(1) activations and
# unit-variance weights, and the data gradient is range-normalised with the max|dz| its producer (epilogue_bwd) reports -- everything
# else (toRGB, the generic conv2d of conv2d_gradfix) runs 'bf16x6'.  Any other value forces that mode everywhere.
# Set with set_conv_precision() or the EG3D_CONV_PRECISION environment variable.
PRECISIONS = {'f32': 0, 'bf16x6': 1, 'bf16x3': 2, 'f16x3': 3, 'f16x1': 4}      # 'f16x1': single product of the high fp16 pieces (include/eg3d_hip.h)
CONV_MODE = os.environ.get('EG3D_CONV_PRECISION', 'auto')
CONV_PRECISION = PRECISIONS['bf16x6' if CONV_MODE == 'auto' else CONV_MODE]


def set_conv_precision(name):
    global CONV_MODE, CONV_PRECISION
    assert name == 'auto' or name in PRECISIONS
    CONV_MODE = name
    CONV_PRECISION = PRECISIONS['bf16x6' if name == 'auto' else name]


MODCONV_OVERRIDE = None      # 'f16x1' inside modconv_override(): every modulated conv on the pre-split kernels runs ONE product of fp16-rounded operands


def modconv_precision() -> str:
    """Arithmetic of the style-modulated convolutions (the dominant kernels) under the current mode."""
    if MODCONV_OVERRIDE is not None and CONV_MODE == 'auto':
        return MODCONV_OVERRIDE
    return 'f16x3' if CONV_MODE == 'auto' else CONV_MODE


@contextlib.contextmanager
def modconv_override(name):
    """Arithmetic class of the reference's own GPU path for the modulated convs of BOTH networks (backbone + super-resolution head): `f16x1` =
    one v_mfma_f32_32x32x16_f16 product of range-normalised fp16-rounded operands, fp32 accumulation, fp32 results -- an 11-bit significand
    per operand, what a TF32 convolution keeps (the reference never disables TF32 on its inversion path: only training/training_loop.py:135-136
    and calc_metrics.py:52-53 do).  Layers still on the loader-split kernel (it has no single-product form) keep three products."""
    global MODCONV_OVERRIDE
    assert name in (None, 'f16x1', 'f16x3')
    prev, MODCONV_OVERRIDE = MODCONV_OVERRIDE, name
    try:
        yield
    finally:
        MODCONV_OVERRIDE = prev


class ActBwdSpec:
    """What EPI_BWD_ACT needs about the layer that produced a data gradient's `xin`: that layer's epilogue constants (saved at its forward)
    and the pre-zeroed reduction targets of its backward (filled by the fused launch)."""
    __slots__ = ('d', 'bias', 'noise', 'noise_nstride', 'noise_strength', 'act', 'alpha', 'gain', 'clamp', 'dbias', 'dd', 'dnoise',
                 'dnoise_nstride', 'dstrength')

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def fill(self, ab):
        tp = lambda t: t.data_ptr() if t is not None else None
        ab.d, ab.bias, ab.noise, ab.noise_strength = tp(self.d), tp(self.bias), tp(self.noise), tp(self.noise_strength)
        ab.noise_nstride, ab.dnoise_nstride = int(self.noise_nstride or 0), int(self.dnoise_nstride or 0)
        ab.act, ab.alpha, ab.gain, ab.clamp = L.ACT_IDS[self.act], float(self.alpha), float(self.gain), float(self.clamp)
        ab.dbias, ab.dd, ab.dnoise, ab.dstrength = tp(self.dbias), tp(self.dd), tp(self.dnoise), tp(self.dstrength)


def conv_igemm(x, wp, Ck, Nc, out, classes, in_stride=1, out_stride=1, in_scale=None, epi=L.EPI_STORE, ksplit=1,
               out_scale=None, bias=None, noise=None, noise_nstride=0, noise_strength=None, act='linear', alpha=0.0, gain=1.0,
               clamp=-1.0, addend=None, xin=None, ds=None, algo_flops=None, precision=None, a_amax=None, a_amax_mul=1.0, out_amax=None,
               act_bwd=None, w_pieces=None, addend_up2_taps=None):
    """Launch eg3d_conv2d_igemm_f32.  x/out/addend/xin: channels_last fp32 [N,C,H,W]; wp: packed weights [Nc, taps*Ck].
    act_bwd (an ActBwdSpec, with epi=EPI_BWD): try EPI_BWD_ACT; returns True when the fused epilogue ran (out then holds the producing
    layer's dz), False when the launch was a plain EPI_BWD."""
    assert is_cl(x) and is_cl(out), 'conv_igemm expects fp32 channels_last CUDA tensors'
    if len(classes) > 4:        # (a stride-3 transposed conv has nine output phases) the classes of a store / atomic launch are independent: four per launch
        assert epi in (L.EPI_STORE, L.EPI_ATOMIC) and ds is None and out_amax is None and act_bwd is None, 'more than four tap classes: plain store launches only'
        for i in range(0, len(classes), 4):
            conv_igemm(x, wp, Ck, Nc, out, classes[i:i + 4], in_stride=in_stride, out_stride=out_stride, in_scale=in_scale, epi=epi, ksplit=ksplit,
                       out_scale=out_scale, algo_flops=None if algo_flops is None else algo_flops * min(4, len(classes) - i) / len(classes),
                       precision=precision, a_amax=a_amax, a_amax_mul=a_amax_mul, w_pieces=w_pieces)
        return False
    p = L.ConvParams()
    n, cx, hi, wi = x.shape
    _, co, ho, wo = out.shape
    p.x, p.w, p.out = x.data_ptr(), wp.data_ptr(), out.data_ptr()
    p.N, p.Hi, p.Wi, p.Ck, p.ldx = n, hi, wi, Ck, cx
    p.Nc, p.w_row = Nc, wp.stride(0)
    p.Ho, p.Wo, p.ldo = ho, wo, co
    p.in_stride, p.out_stride = in_stride, out_stride
    p.ncls = len(classes)
    for i, c in enumerate(classes):
        p.cls[i] = c
    p.in_scale = in_scale.data_ptr() if in_scale is not None else None
    p.epi, p.ksplit = epi, ksplit
    p.out_scale = out_scale.data_ptr() if out_scale is not None else None
    p.bias = bias.data_ptr() if bias is not None else None
    p.noise = noise.data_ptr() if noise is not None else None
    p.noise_nstride = noise_nstride
    p.noise_strength = noise_strength.data_ptr() if noise_strength is not None else None
    p.act, p.alpha, p.gain, p.clamp = L.ACT_IDS[act], float(alpha), float(gain), float(clamp)
    p.addend = addend.data_ptr() if addend is not None else None
    if addend_up2_taps is not None:             # addend is the half-resolution skip image, up-sampled inside the epilogue
        p.addend_up2 = 1
        p.addend_taps[:] = [float(t) for t in addend_up2_taps]
    p.xin = xin.data_ptr() if xin is not None else None
    p.ds = ds.data_ptr() if ds is not None else None
    p.precision = CONV_PRECISION if precision is None else PRECISIONS[precision]
    p.a_amax = a_amax.data_ptr() if a_amax is not None else None
    p.a_amax_mul = float(a_amax_mul)
    p.ds_replicas = ds.shape[0] if (ds is not None and ds.dim() == 3) else 1
    p.out_amax = out_amax.data_ptr() if out_amax is not None else None
    if w_pieces is not None and p.precision == PRECISIONS['f16x3']:      # the packed weights' pre-split image (split_weight_pieces): same indexing
        assert w_pieces.shape == wp.shape and w_pieces.stride() == wp.stride()
        p.w, p.w_presplit = w_pieces.data_ptr(), 1
    fused_act = False
    if act_bwd is not None and epi == L.EPI_BWD:
        p.epi = L.EPI_BWD_ACT
        act_bwd.fill(p.act_bwd)
        fused_act = bool(L.lib().eg3d_conv2d_igemm_act_bwd_ok(C.byref(p)))
        if not fused_act:
            p.epi = L.EPI_BWD
            p.act_bwd = L.ActBwd()
    prof = PROFILER
    if prof is not None:
        cfg = L.lib().eg3d_conv2d_igemm_config(C.byref(p))
        if prof.only_config is not None and cfg != prof.only_config:
            prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * Ck * Nc * sum(n * c.Ha * c.Wa * c.ntaps for c in classes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().eg3d_conv2d_igemm_f32(C.byref(p), L.stream_ptr()), 'conv2d_igemm_f32')
    if prof is not None:
        e1.record()
        prof.records.append(((cfg, int(p.precision)), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=n, Hi=hi, Wi=wi, Ck=Ck, Nc=Nc, Ho=ho, Wo=wo, taps=[c.ntaps for c in classes], epi=epi, ksplit=ksplit,
                                  in_stride=in_stride, out_stride=out_stride, prec=p.precision))
    return fused_act if act_bwd is not None else out


# ------------------------------------------------------------------------------------------------- pre-split convolution (csrc/conv_v2.hip)
def absmax(x, out=None):
    """Device scalar max|x| (eg3d_absmax); `out`: a pre-zeroed 1-element tensor to accumulate into."""
    if out is None:
        out = zeros((1,), x.device)
    L.check(L.lib().eg3d_absmax(L.ptr(x), x.numel(), L.ptr(out), L.stream_ptr()), 'absmax')
    return out


def tag_amax(t, amax):
    """Attach the device scalar max|t| its producer kernel reported to a tensor (read back by amax_of)."""
    t._eg3d_amax = amax
    return t


def amax_of(t):
    """Device scalar max|t|: the producer's report when the tensor carries one, else one reduction pass (eg3d_absmax)."""
    a = getattr(t, '_eg3d_amax', None)
    return a if a is not None else absmax(t)


class SplitImage:
    """Two-piece fp16 image of an operand + the power-of-two scale it was written with (device scalar)."""
    __slots__ = ('data', 'scale', 'shape')

    def __init__(self, data, scale, shape):
        self.data, self.scale, self.shape = data, scale, shape


def split_activation(x, x_amax, in_scale=None, s_amax=None, C=None):
    """fp32 channels_last [N,C,H,W] (times in_scale[n,c]) -> SplitImage [N][2][C/8][H][W][8] fp16, range-normalised with the device scalars
    x_amax = max|x| and s_amax = max|in_scale|."""
    assert is_cl(x)
    n, cx, h, w = x.shape
    C_ = cx if C is None else C
    img = torch.empty((n * h * w * C_ * 2,), dtype=torch.float16, device=x.device)
    scale = torch.empty((1,), dtype=torch.float32, device=x.device)
    L.check(L.lib().eg3d_split_activation(L.ptr(x), L.ptr(in_scale), L.ptr(x_amax), L.ptr(s_amax), L.ptr(img), L.ptr(scale), n, h, w, C_, cx,
                                          L.stream_ptr()), 'split_activation')
    return SplitImage(img, scale, (n, C_, h, w))


def split_weight(wp, O, I, T):
    """Packed fp32 weights [O, T*I] (forward or adjoint image) -> SplitImage [T][I/16][2][2][O][8] fp16 (unscaled low piece)."""
    amax = absmax(wp)
    img = torch.empty((O * T * I * 2,), dtype=torch.float16, device=wp.device)
    scale = torch.empty((1,), dtype=torch.float32, device=wp.device)
    L.check(L.lib().eg3d_split_weight(L.ptr(wp), L.ptr(amax), L.ptr(img), L.ptr(scale), O, I, T, wp.stride(0), L.stream_ptr()), 'split_weight')
    return SplitImage(img, scale, (O, I, T))


def split_weights_batched(mats):
    """[(packed fp32 matrix [O, T*I], O, I, T), ...] -> [SplitImage, ...] from two launches per batch of SPLIT_W_BATCH_MAX (eg3d_split_weights_batched)."""
    outs = []
    for b0 in range(0, len(mats), L.SPLIT_W_BATCH_MAX):
        grp = mats[b0:b0 + L.SPLIT_W_BATCH_MAX]
        dev = grp[0][0].device
        amax = zeros((len(grp),), dev)
        scales = torch.empty((len(grp),), dtype=torch.float32, device=dev)
        b = L.SplitWBatch()
        b.n = len(grp)
        for k, (wp, O, I, T) in enumerate(grp):
            assert wp.is_contiguous() and wp.shape == (O, T * I)
            img = torch.empty((O * T * I * 2,), dtype=torch.float16, device=dev)
            it = b.items[k]
            it.w, it.image, it.scale_out, it.amax = wp.data_ptr(), img.data_ptr(), scales[k:k + 1].data_ptr(), amax[k:k + 1].data_ptr()
            it.O, it.I, it.T, it.w_row = O, I, T, T * I
            outs.append(SplitImage(img, scales[k:k + 1], (O, I, T)))
        L.check(L.lib().eg3d_split_weights_batched(C.byref(b), L.stream_ptr()), 'split_weights_batched')
    return outs


def _conv_v2_params(a, w, out, classes, out_stride, epi, out_scale, bias, noise, noise_nstride, noise_strength, act, alpha, gain, clamp,
                    addend, xin, ds, out_amax):
    p = L.ConvV2Params()
    n, ck, hi, wi = a.shape
    nc, _, wtaps = w.shape
    _, co, ho, wo = out.shape
    p.a, p.w, p.a_scale, p.w_scale, p.out = a.data.data_ptr(), w.data.data_ptr(), a.scale.data_ptr(), w.scale.data_ptr(), out.data_ptr()
    p.N, p.Hi, p.Wi, p.Ck, p.Nc, p.wtaps = n, hi, wi, ck, nc, wtaps
    p.Ho, p.Wo, p.ldo, p.in_stride, p.out_stride = ho, wo, co, 1, out_stride
    p.ncls = len(classes)
    for i, c in enumerate(classes):
        p.cls[i] = c
    p.epi = epi
    p.out_scale = out_scale.data_ptr() if out_scale is not None else None
    p.bias = bias.data_ptr() if bias is not None else None
    p.noise = noise.data_ptr() if noise is not None else None
    p.noise_nstride = noise_nstride
    p.noise_strength = noise_strength.data_ptr() if noise_strength is not None else None
    p.act, p.alpha, p.gain, p.clamp = L.ACT_IDS[act], float(alpha), float(gain), float(clamp)
    p.addend = addend.data_ptr() if addend is not None else None
    p.xin = xin.data_ptr() if xin is not None else None
    p.ds = ds.data_ptr() if ds is not None else None
    p.out_amax = out_amax.data_ptr() if out_amax is not None else None
    return p


def conv_v2_tiles(Nc, classes, N=1):
    """256-cell x 128-channel tiles of a launch of the pre-split kernel."""
    return sum(N * -(-c.Ha // 8) * -(-c.Wa // 32) for c in classes) * (Nc // 128)


def conv_v2_geometry_ok(Ck, Nc, classes):
    if Ck % 16 or Nc % 128 or CONV_MODE != 'auto':
        return False
    for c in classes:
        if c.ntaps not in (9, 4, 2, 1):
            return False
        dys, dxs = [c.dy[t] for t in range(c.ntaps)], [c.dx[t] for t in range(c.ntaps)]
        if max(dys) - min(dys) > 2 or max(dxs) - min(dxs) > 2:
            return False
    return True


def conv_v2_rows(Ck, Nc, classes, N=1):
    """Patch rows (8 | 4 | 2) the pre-split kernel should run this launch with, 0 = not a launch for it.  8 x 32-cell patches when they fill the
    chip (>= V2_MIN_TILES workgroups); nine-tap classes whose 8-row grid does not but whose 4-row grid does (128^2 x 256: 128 -> 256
    workgroups) take the half-height patch -- fused epilogues intact, no split-K."""
    if not conv_v2_geometry_ok(Ck, Nc, classes):
        return 0
    tiles8 = conv_v2_tiles(Nc, classes, N)
    half_ok = V2_HALF and all(c.ntaps == 9 for c in classes)
    tiles4 = sum(N * -(-c.Ha // 4) * -(-c.Wa // 32) for c in classes) * (Nc // 128)
    if half_ok and tiles8 < V2_HALF_BELOW and tiles4 >= V2_MIN_TILES:
        return 4
    if tiles8 >= V2_MIN_TILES:
        return 8
    # 64^2 x 512 (64 workgroups of 8 rows): 2 x 32-cell patches give 256 (opt-in: V2_QUARTER)
    tiles2 = sum(N * -(-c.Ha // 2) * -(-c.Wa // 32) for c in classes) * (Nc // 128)
    return 2 if (half_ok and V2_QUARTER and tiles2 >= V2_MIN_TILES) else 0


def conv_v2_supported(Ck, Nc, classes, N=1):
    """Geometry the pre-split kernel takes (see eg3d_conv2d_v2_supported) AND enough tiles to fill the chip (conv_v2_rows)."""
    return conv_v2_rows(Ck, Nc, classes, N) != 0


def conv_v2_ksplit(Ck, Nc, classes, N=1):
    """Split-K factor for a 3x3 layer whose grid cannot fill the chip with 256 x 128 tiles (128^2 x 256: 128 tiles, 64^2 x 512: 64):
    slices of >= V2_KS_MIN_CHUNKS 16-channel chunks, aiming at two workgroups per CU.  0 = leave the layer to the loader-split kernel."""
    if not (USE_V2 and V2_SPLITK) or len(classes) != 1 or classes[0].ntaps != 9 or not conv_v2_geometry_ok(Ck, Nc, classes):
        return 0
    tiles = conv_v2_tiles(Nc, classes, N)
    if tiles >= V2_MIN_TILES or tiles < V2_KS_MIN_TILES:
        return 0
    ks = min(-(-V2_KS_TARGET // tiles), (Ck // 16) // V2_KS_MIN_CHUNKS)
    return ks if ks >= 2 else 0


V2_MIN_TILES = 256
USE_V2 = os.environ.get('EG3D_CONV_V2', '1') != '0'
V2_CONVT = False          # (was EG3D_V2_CONVT: the up layers on tap classes of the pre-split kernel -- superseded by conv_up2; tests set the attribute)
# 4 x 32-cell patches for under-filled 3x3 grids (conv_v2_rows): 128^2 x 256: 107 -> 91 us, 256^2 x 128: 84 -> 67 us per launch.  (Off in the
# first half of round 3: with the fused epilogues a few hundred of 8 M output elements per launch came out wrong on full-size layers --
# a miscompile of the scalar epilogue arithmetic by the SLP vectoriser, see the Makefile; tests/test_gpu_ops.py::test_conv_v2_half_patch_full_size.)
# 4-row launches that give every CU at most one workgroup (128^2 x 256, 256^2 x 128 at N = 1) as eight-wave workgroups whose two halves split the contraction
# (csrc/conv_v2.hip KH = 2): two waves per SIMD instead of one
V2_KHALVES = os.environ.get('EG3D_V2_KHALVES', '1') != '0'
V2_KHALVES_MAX_TILES = 256
V2_HALF = os.environ.get('EG3D_V2_HALF', '1') != '0'             # half-height (4 x 32) patches for nine-tap launches ...
# quarter-height (2 x 32) patches where not even the 4-row grid fills the chip (64^2 x 512: 64 -> 256 workgroups): OFF by default.  Stand-alone
# (weights warm in L2) 74 us against 99 + 23 (4-way split-K launch of the loader-split kernel + its finishing pass); inside the step, where every
# workgroup streams its 2.4 MB weight slice from HBM / MALL, 109 us + the 12 us operand split: no gain (194.3 vs 194.3 steps/s, A/B in one session)
V2_QUARTER = False        # (was EG3D_V2_QUARTER; lost its A/B twice -- tests set the attribute)
V2_HALF_BELOW = 512      # ... when the 8-row grid has fewer workgroups than this (256^2 x 128: 84 -> 67 us, 128^2 x 256: 107 -> 91 us)
# split-K launches of the pre-split kernel for under-filled 3x3 grids: OFF by default.  Measured at N = 1 (MI355X): 128^2 x 256 118 -> 81 us,
# 64^2 x 512 103 -> 82 us per launch, but the operand split pass (7 us), the zero fill and the finishing pass (2 x 10 us; the loader-split
# kernel's fused epilogue needs neither on the 128^2 layer) eat it: -1.2 % per step.  Batched runs do not need it (the grids fill).
V2_SPLITK = False         # (was EG3D_V2_SPLITK; -1.2 % twice -- tests set the attribute)
V2_KS_TARGET = 512        # workgroups a split launch aims for (2 per CU)
V2_KS_MIN_TILES = 32      # 32^2 x 512 (16 tiles): no gain over the loader-split kernel (46.8 vs 46.4 us)
V2_KS_MIN_CHUNKS = 2


def conv_v2(a: SplitImage, w: SplitImage, out, classes, out_stride=1, epi=L.EPI_STORE, out_scale=None, bias=None, noise=None, noise_nstride=0,
            noise_strength=None, act='linear', alpha=0.0, gain=1.0, clamp=-1.0, addend=None, xin=None, ds=None, out_amax=None, algo_flops=None,
            act_bwd=None, products=3, ksplit=1, patch_rows=None, rgb_head=None, rgb_head_ran=None):
    """Launch eg3d_conv2d_v2 (operands prepared by split_activation / split_weight).  act_bwd (ActBwdSpec, with epi=EPI_BWD): EPI_BWD_ACT when
    the kernel takes it -- returns True if the fused epilogue ran, False for a plain EPI_BWD.
    rgb_head (EPI_FWD, 128 output channels): (w4 [4, >= Nc] packed weight rows, styles [N, Nc], bias4 [4] | None, y4 [N,4,H,W] channels_last, clamp[, 3 = the
    fourth row is padding]) --
    the 1x1 layer that reads `out` next, evaluated in this launch's epilogue (eg3d_conv_v2_params::rgb_out); rgb_head_ran (a list): receives True / False =
    the library took / refused the head (refused: the launch runs without it)."""
    assert is_cl(out)
    p = _conv_v2_params(a, w, out, classes, out_stride, epi, out_scale, bias, noise, noise_nstride, noise_strength, act, alpha, gain, clamp,
                        addend, xin, ds, out_amax)
    p.products = int(products)
    p.ksplit = int(ksplit)
    assert ksplit <= 1 or epi == L.EPI_ATOMIC
    if patch_rows is None:          # the caller did not plan: 4-row patches where they fill the chip and 8-row ones do not
        patch_rows = conv_v2_rows(p.Ck, p.Nc, classes, p.N) if epi != L.EPI_ATOMIC else 8
    p.patch_rows = patch_rows if patch_rows in (4, 2) else 8
    if (V2_KHALVES and p.patch_rows == 4 and epi != L.EPI_ATOMIC and ksplit <= 1 and rgb_head is None and (p.Ck // 16) % 2 == 0 and p.Ck >= 64
            and sum(p.N * -(-c.Ha // 4) * -(-c.Wa // 32) for c in classes) * (p.Nc // 128) <= V2_KHALVES_MAX_TILES):
        # one workgroup per CU: eight waves, the contraction split over the two four-wave halves inside the workgroup (KH = 2 of conv_v2_kernel)
        p.ksplit = 2
    if rgb_head is not None:
        w4, s4, b4, y4, rclamp = rgb_head[:5]
        p.rgb_nout = int(rgb_head[5]) if len(rgb_head) > 5 else 4
        assert epi == L.EPI_FWD and is_cl(y4) and y4.shape[1] == 4 and w4.shape[0] == 4 and w4.stride(1) == 1 and s4.is_contiguous()
        p.rgb_w, p.rgb_s, p.rgb_bias, p.rgb_out = w4.data_ptr(), s4.data_ptr(), (b4.data_ptr() if b4 is not None else None), y4.data_ptr()
        p.rgb_clamp, p.rgb_ldw = float(rclamp), w4.stride(0)
        if not L.lib().eg3d_conv2d_v2_supported(C.byref(p)):        # (alignment of the head's operands, patch height ...): this launch without the head, the caller launches the 1x1 layer itself
            p.rgb_w = p.rgb_s = p.rgb_bias = p.rgb_out = None
            p.rgb_nout = 0
            rgb_head = None
        if rgb_head_ran is not None:
            rgb_head_ran.append(rgb_head is not None)
    fused_act = False
    if act_bwd is not None and epi == L.EPI_BWD and xin is not None:
        p.epi = L.EPI_BWD_ACT
        act_bwd.fill(p.act_bwd)
        fused_act = bool(L.lib().eg3d_conv2d_v2_supported(C.byref(p))) and all(
            t is None or t.data_ptr() % 16 == 0 for t in (act_bwd.d, act_bwd.bias))
        if not fused_act:
            p.epi = L.EPI_BWD
            p.act_bwd = L.ActBwd()
    prof = PROFILER
    cfg_id = {8: V2_CONFIG, 4: V2H_CONFIG, 2: V2Q_CONFIG}[int(p.patch_rows)]   # the patch heights are different instantiations: separate profiler records
    if rgb_head is not None:
        cfg_id = V2RGB_CONFIG       # ... and so is the one with the 1x1 head (conv_v2_kernel<9,*,false,4,true>)
    if prof is not None and prof.only_config is not None and prof.only_config != cfg_id:
        prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * p.Ck * p.Nc * sum(p.N * c.Ha * c.Wa * c.ntaps for c in classes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().eg3d_conv2d_v2(C.byref(p), L.stream_ptr()), 'conv2d_v2')
    if prof is not None:
        e1.record()
        prof.records.append(((cfg_id, PRECISIONS['f16x3']), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=p.N, Hi=p.Hi, Wi=p.Wi, Ck=p.Ck, Nc=p.Nc, Ho=p.Ho, Wo=p.Wo, taps=[c.ntaps for c in classes], epi=epi, ksplit=int(ksplit),
                                  in_stride=1, out_stride=out_stride, prec=3, v2=True, patch_rows=int(p.patch_rows)))
    return fused_act if act_bwd is not None else out


# ------------------------------------------------------------------------------------------------- wave-split pre-split convolution (csrc/conv_v3.hip)
V3_CONFIG = 11       # profiler id of conv_v3_kernel
USE_V3 = os.environ.get('EG3D_CONV_V3', '1') != '0'
V3_MAX_TILES8 = 128       # 256-cell x 128-channel tiles below which a 3x3 launch goes to the wave-split kernel
V3_MIN_CELLS = 1024        # class grids (all images) from this many cells (32^2) -- below, launches are latency-bound


def conv_v3_plan(Ck, Nc, classes, N=1):
    """(patch_rows, waves) for eg3d_conv2d_v3, or None when the launch is not one for it: nine-tap stride-1 classes whose 256 x 128 tiling
    leaves the chip under-filled.  128-cell tiles with the contraction over four waves when they give >= 192 workgroups, else 64-cell tiles
    over eight waves (twice the workgroups, half the steps per wave)."""
    if not USE_V3 or CONV_MODE != 'auto' or Ck % 16 or Nc % 64 or not (1 <= len(classes) <= 4):
        return None
    for c in classes:
        if c.ntaps != 9:
            return None
        dys, dxs = [c.dy[t] for t in range(9)], [c.dx[t] for t in range(9)]
        if max(dys) - min(dys) > 2 or max(dxs) - min(dxs) > 2:
            return None
    cells = sum(N * c.Ha * c.Wa for c in classes)
    tiles8 = sum(N * -(-c.Ha // 8) * -(-c.Wa // 32) for c in classes) * -(-Nc // 128)
    if cells < V3_MIN_CELLS or tiles8 >= V3_MAX_TILES8:
        return None
    wg4 = sum(N * -(-c.Ha // 4) * -(-c.Wa // 32) for c in classes) * (Nc // 64)
    return (4, 4) if wg4 >= 192 else (2, 8)


def conv_v3(a: SplitImage, w: SplitImage, out, classes, plan=None, out_stride=1, epi=L.EPI_STORE, out_scale=None, bias=None, noise=None, noise_nstride=0,
            noise_strength=None, act='linear', alpha=0.0, gain=1.0, clamp=-1.0, addend=None, xin=None, ds=None, out_amax=None, algo_flops=None,
            act_bwd=None, products=3):
    """Launch eg3d_conv2d_v3 (operands as for conv_v2).  plan = (patch_rows, waves).  act_bwd (ActBwdSpec, with epi=EPI_BWD): EPI_BWD_ACT when the
    kernel takes it -- returns True if the fused epilogue ran, False for a plain EPI_BWD."""
    assert is_cl(out)
    p = _conv_v2_params(a, w, out, classes, out_stride, epi, out_scale, bias, noise, noise_nstride, noise_strength, act, alpha, gain, clamp,
                        addend, xin, ds, out_amax)
    rows, waves = plan if plan is not None else (4, 4)
    p.products, p.patch_rows, p.ksplit = int(products), int(rows), int(waves)
    fused_act = False
    if act_bwd is not None and epi == L.EPI_BWD and xin is not None:
        p.epi = L.EPI_BWD_ACT
        act_bwd.fill(p.act_bwd)
        fused_act = bool(L.lib().eg3d_conv2d_v3_supported(C.byref(p))) and all(
            t is None or t.data_ptr() % 16 == 0 for t in (act_bwd.d, act_bwd.bias))
        if not fused_act:
            p.epi = L.EPI_BWD
            p.act_bwd = L.ActBwd()
    prof = PROFILER
    if prof is not None and prof.only_config is not None and prof.only_config != V3_CONFIG:
        prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * p.Ck * p.Nc * sum(p.N * c.Ha * c.Wa * c.ntaps for c in classes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().eg3d_conv2d_v3(C.byref(p), L.stream_ptr()), 'conv2d_v3')
    if prof is not None:
        e1.record()
        prof.records.append(((V3_CONFIG, PRECISIONS['f16x3']), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=p.N, Hi=p.Hi, Wi=p.Wi, Ck=p.Ck, Nc=p.Nc, Ho=p.Ho, Wo=p.Wo, taps=[c.ntaps for c in classes], epi=epi, ksplit=int(waves),
                                  in_stride=1, out_stride=out_stride, prec=3, v3=True, patch_rows=int(rows)))
    return fused_act if act_bwd is not None else out


RGB_HEAD = os.environ.get('EG3D_RGB_HEAD', '1') != '0'       # the SR head's last toRGB evaluated in conv1's forward epilogue (eg3d_conv_v2_params::rgb_out)
CONV_WS = os.environ.get('EG3D_CONV_WS', '1') != '0'
CONV_WS_MAX_CELLS = 256
# ... NOT when the weights carry gradients (pivotal tuning): measured in round 6 (A/B twice in one session, graph-replayed C4 step) 6.22 / 6.21 ms without,
# 6.29 / 6.26 ms with -- re-splitting six more 9.4 MB weight tensors per step (forward + adjoint images) costs more than the twelve 19.7 us launches save
CONV_WS_TRAINABLE = False
CONV_WS_S2 = os.environ.get('EG3D_CONV_WS_S2', '1') != '0'      # ... and its stride-2 adjoint form for the data gradients of the 8^2 .. 32^2 up layers
WS_CONFIG = 12


def _conv_ws_params(x, w: SplitImage, out, cls, in_scale, x_amax, x_amax_mul, products, in_stride=1):
    p = L.ConvWsParams()
    n, cx, hx, wx = x.shape
    _, co, ho, wo = out.shape
    assert cls.ntaps == 9 and (cls.Ha, cls.Wa, cls.out_py, cls.out_px) == (ho, wo, 0, 0) and (in_stride == 2 or (hx, wx) == (ho, wo))
    p.x, p.in_scale, p.x_amax, p.x_amax_mul = x.data_ptr(), (in_scale.data_ptr() if in_scale is not None else None), x_amax.data_ptr(), float(x_amax_mul)
    p.w, p.w_scale, p.out = w.data.data_ptr(), w.scale.data_ptr(), out.data_ptr()
    O, I, T = w.shape
    p.N, p.H, p.W, p.Ck, p.ldx = n, ho, wo, I, cx
    p.Nc, p.ldo, p.wtaps = O, co, T
    for t in range(9):
        p.dy[t], p.dx[t], p.wtap[t] = cls.dy[t], cls.dx[t], cls.wtap[t]
    p.products = int(products)
    p.in_stride, p.Hx, p.Wx = int(in_stride), hx, wx
    return p


def conv_ws_ok(Ck, Nc, classes, N, H, W, in_stride=1):
    """A split (atomic) 3x3 launch on the weight-streaming kernel (csrc/conv_ws.hip): the stride-1 layers of the 4^2 .. 16^2 blocks at one image per GPU;
    in_stride 2: the stride-2 adjoint (data gradient of an up layer) with H x W <= 256 output cells."""
    if not (USE_V2 and CONV_WS) or CONV_MODE != 'auto' or len(classes) != 1 or classes[0].ntaps != 9 or Ck % 16 or Nc % 32 or W > 32:
        return False
    c = classes[0]
    if (c.Ha, c.Wa, c.out_py, c.out_px) != (H, W, 0, 0):
        return False
    if in_stride == 2:
        if not CONV_WS_S2 or any(not (0 <= c.dy[t] <= 2 and 0 <= c.dx[t] <= 2) for t in range(9)) or H * W > 256 or N != 1:
            return False
        return _conv_ws_geometry_ok(Ck, Nc, N, H, W, c, 2, 2 * H + 1, 2 * W + 1, 1)
    if any(abs(c.dy[t]) > 1 or abs(c.dx[t]) > 1 for t in range(9)):
        return False
    # the routing threshold, then the library's own geometry check (e.g. a 64 x 4 grid passes the cell count but not the kernel's halo-slot limit): a launch the
    # kernel refuses must fall back to the split-K implicit GEMM here, not raise from L.check later
    return N * H * W <= CONV_WS_MAX_CELLS and _conv_ws_geometry_ok(Ck, Nc, N, H, W, c, 1, H, W, 1)


def _conv_ws_geometry_ok(Ck, Nc, N, H, W, c, in_stride, Hx, Wx, out_stride):
    """eg3d_conv2d_ws_supported on the geometry alone (it reads no pointer)."""
    p = L.ConvWsParams()
    p.N, p.H, p.W, p.Ck, p.ldx, p.Nc, p.ldo, p.wtaps = N, H, W, Ck, Ck, Nc, Nc, 9
    for t in range(9):
        p.dy[t], p.dx[t], p.wtap[t] = (c.dy[t], c.dx[t], c.wtap[t]) if c is not None else (0, 0, t)
    p.products, p.in_stride, p.Hx, p.Wx, p.out_stride = 3, in_stride, Hx, Wx, out_stride
    return bool(L.lib().eg3d_conv2d_ws_supported(C.byref(p)))


def conv_ws(x, w: SplitImage, out, classes, in_scale=None, x_amax=None, x_amax_mul=1.0, products=3, algo_flops=None, in_stride=1):
    """Launch eg3d_conv2d_ws: out (pre-zeroed, channels_last fp32) += conv(x * in_scale, W) over the nine taps of classes[0]; w: split_weight image.
    in_stride 2: the stride-2 adjoint form (x: the FIR-adjointed gradient of an up layer, out: its data gradient before the style scale)."""
    assert is_cl(x) and is_cl(out) and x.dtype == torch.float32
    p = _conv_ws_params(x, w, out, classes[0], in_scale, x_amax if x_amax is not None else absmax(x), x_amax_mul, products, in_stride)
    prof = PROFILER
    if prof is not None and prof.only_config is not None and prof.only_config != WS_CONFIG:
        prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * p.Ck * p.Nc * p.N * p.H * p.W * 9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().eg3d_conv2d_ws(C.byref(p), L.stream_ptr()), 'conv2d_ws')
    if prof is not None:
        e1.record()
        prof.records.append(((WS_CONFIG, PRECISIONS['f16x3']), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=p.N, Hi=p.Hx, Wi=p.Wx, Ck=p.Ck, Nc=p.Nc, Ho=p.H, Wo=p.W, taps=[9], epi=L.EPI_ATOMIC, ksplit=p.Ck // 16,
                                  in_stride=int(in_stride), out_stride=1, prec=3, ws=True))
    return out


CONV_WS_UP = os.environ.get('EG3D_CONV_WS_UP', '1') != '0'      # ... and its transposed form for the forward of the 8^2 block's up layer
CONV_WS_UP_MAX_CELLS = 32      # cells per output parity routed there: 4^2 -> 9^2 (25 cells) 18.9 -> 12.4 us; 8^2 -> 17^2 (81 cells: three row tiles x four parity sets per wave, 1.2 M output atomics) 18.0 -> 23.1, stays on the split-K implicit GEMM


def conv_ws_up_ok(Ck, Nc, N, H, W, max_cells=None):
    """Forward of an up layer (stride-2 transposed 3x3 conv, split-K accumulation into a zeroed buffer) on the weight-streaming kernel: 4^2 / 8^2
    input cells at one image per GPU.  max_cells: cells per output parity the caller accepts (default: the routing threshold; the kernel takes 96)."""
    return bool(USE_V2 and CONV_WS and CONV_WS_UP and CONV_MODE == 'auto' and Ck % 16 == 0 and Nc % 32 == 0 and N == 1 and (H + 1) * (W + 1) <= min(96, max_cells or CONV_WS_UP_MAX_CELLS)
                and W <= 30 and _conv_ws_geometry_ok(Ck, Nc, N, H, W, None, 1, H, W, 2))


def conv_ws_up(x, w: SplitImage, out, in_scale=None, x_amax=None, x_amax_mul=1.0, products=3, algo_flops=None):
    """Launch eg3d_conv2d_ws with out_stride 2: out [N,Co,2H+1,2W+1] (pre-zeroed, channels_last) += conv_transpose2d(x * in_scale, W, stride 2);
    w: the FORWARD split weight image (tap index 3 ky + kx)."""
    assert is_cl(x) and is_cl(out) and x.dtype == torch.float32
    p = L.ConvWsParams()
    n, cx, hx, wx = x.shape
    _, co, ho, wo = out.shape
    assert (ho, wo) == (2 * hx + 1, 2 * wx + 1)
    xa = x_amax if x_amax is not None else absmax(x)
    p.x, p.in_scale, p.x_amax, p.x_amax_mul = x.data_ptr(), (in_scale.data_ptr() if in_scale is not None else None), xa.data_ptr(), float(x_amax_mul)
    p.w, p.w_scale, p.out = w.data.data_ptr(), w.scale.data_ptr(), out.data_ptr()
    O, I, T = w.shape
    p.N, p.H, p.W, p.Ck, p.ldx = n, hx, wx, I, cx
    p.Nc, p.ldo, p.wtaps = O, co, T
    for t in range(9):
        p.dy[t], p.dx[t], p.wtap[t] = 0, 0, t
    p.products, p.in_stride, p.out_stride = int(products), 1, 2
    prof = PROFILER
    if prof is not None and prof.only_config is not None and prof.only_config != WS_CONFIG:
        prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * p.Ck * p.Nc * p.N * p.H * p.W * 9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().eg3d_conv2d_ws(C.byref(p), L.stream_ptr()), 'conv2d_ws (transposed)')
    if prof is not None:
        e1.record()
        prof.records.append(((WS_CONFIG, PRECISIONS['f16x3']), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=p.N, Hi=p.H, Wi=p.W, Ck=p.Ck, Nc=p.Nc, Ho=ho, Wo=wo, taps=[4, 2, 2, 1], epi=L.EPI_ATOMIC, ksplit=p.Ck // 16,
                                  in_stride=1, out_stride=2, prec=3, ws=True))
    return out


def fir44_adjoint_split(dz, dz_amax, gain=4.0):
    """FIR adjoint of an up layer + operand split in one pass (eg3d_fir44_adjoint_split): dz [N,C,2Hi,2Wi] channels_last ->
    SplitImage of the four parity images of G = upfirdn2d(dz, [1,3,3,1]^2 / 64, pad 2, gain), shape (N, C, Hi + 1, Wi + 1) per parity."""
    assert is_cl(dz)
    n, c, ho, wo = dz.shape
    assert ho % 2 == 0 and wo % 2 == 0 and c % 8 == 0
    hi, wi = ho // 2, wo // 2
    img = torch.empty((int(L.lib().eg3d_fir44_adjoint_split_bytes(n, hi, wi, c)) // 2,), dtype=torch.float16, device=dz.device)
    scale = torch.empty((1,), dtype=torch.float32, device=dz.device)
    L.check(L.lib().eg3d_fir44_adjoint_split(L.ptr(dz), L.ptr(dz_amax), L.ptr(img), L.ptr(scale), n, hi, wi, c, c, float(gain), L.stream_ptr()), 'fir44_adjoint_split')
    return SplitImage(img, scale, (n, c, hi + 1, wi + 1))


V2_S2ADJ = os.environ.get('EG3D_V2_S2ADJ', '1') != '0'
S2ADJ_MIN_TILES = 256     # 128 tiles (257^2 x 128 -> 128^2 x 256): 79 us vs 65-70 on the loader-split kernel
S2ADJ_CONFIG = 7


def conv_s2adj_ok(Ck, Nc, Hi, Wi, N=1):
    """The up layers' data gradient on the parity-split kernel (csrc/conv_v2_s2adj.hip): geometry + enough 256 x 128 tiles."""
    if not (USE_V2 and V2_S2ADJ) or CONV_MODE != 'auto' or Ck % 64 or Nc % 128:
        return False
    return N * -(-Hi // 8) * -(-Wi // 32) * (Nc // 128) >= S2ADJ_MIN_TILES


V3_S2ADJ = os.environ.get('EG3D_V3_S2ADJ', '1') != '0'
V3_S2ADJ_MIN_CELLS = 4096


def conv_v3_s2adj_ok(Ck, Nc, Hi, Wi, N=1):
    """The up layers' data gradient on the wave-split form of the parity-split kernel (conv_v3_s2adj_kernel): the grids conv_s2adj_ok turns down
    for want of 256 x 128 tiles, from 32^2 cells."""
    if not (USE_V3 and V3_S2ADJ) or CONV_MODE != 'auto' or Ck % 16 or Nc % 64 or conv_s2adj_ok(Ck, Nc, Hi, Wi, N):
        return False
    return N * Hi * Wi >= V3_S2ADJ_MIN_CELLS


def conv_v2_s2adj(a: SplitImage, w: SplitImage, out, classes, epi=L.EPI_STORE, out_scale=None, addend=None, xin=None, ds=None, out_amax=None, algo_flops=None,
                  act_bwd=None, products=3, ksplit=1, v3=False):
    """Launch eg3d_conv2d_v2_s2adj (v3: eg3d_conv2d_v3_s2adj, the wave-split form): `a` = parity-split image of G (fir44_adjoint_split),
    `classes` = classes_convT_adjoint(...).  Returns like conv_v2."""
    assert is_cl(out) and len(classes) == 1
    fn_ok, fn, nm, cfg_id = ((L.lib().eg3d_conv2d_v3_s2adj_supported, L.lib().eg3d_conv2d_v3_s2adj, 'conv2d_v3_s2adj', V3_CONFIG) if v3 else
                             (L.lib().eg3d_conv2d_v2_s2adj_supported, L.lib().eg3d_conv2d_v2_s2adj, 'conv2d_v2_s2adj', S2ADJ_CONFIG))
    p = _conv_v2_params(a, w, out, classes, 1, epi, out_scale, None, None, 0, None, 'linear', 0.0, 1.0, -1.0, addend, xin, ds, out_amax)
    p.in_stride = 2
    p.products, p.ksplit = int(products), int(ksplit)
    fused_act = False
    if act_bwd is not None and epi == L.EPI_BWD and xin is not None:
        p.epi = L.EPI_BWD_ACT
        act_bwd.fill(p.act_bwd)
        fused_act = bool(fn_ok(C.byref(p))) and all(t is None or t.data_ptr() % 16 == 0 for t in (act_bwd.d, act_bwd.bias))
        if not fused_act:
            p.epi = L.EPI_BWD
            p.act_bwd = L.ActBwd()
    prof = PROFILER
    if prof is not None and prof.only_config is not None and prof.only_config != cfg_id:
        prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * p.Ck * p.Nc * p.N * classes[0].Ha * classes[0].Wa * 9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(fn(C.byref(p), L.stream_ptr()), nm)
    if prof is not None:
        e1.record()
        prof.records.append(((cfg_id, PRECISIONS['f16x3']), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=p.N, Hi=2 * p.Hi - 1, Wi=2 * p.Wi - 1, Ck=p.Ck, Nc=p.Nc, Ho=p.Ho, Wo=p.Wo, taps=[9], epi=epi, ksplit=int(ksplit), in_stride=2,
                                  out_stride=1, prec=3, v2=True))
    return fused_act if act_bwd is not None else out


V2_CONFIG = 5        # "tile configuration" id of the pre-split kernel in profiler records (eg3d_conv2d_igemm_config returns 0..4)
UP2_CONFIG = 6       # ... of the fused-parity transposed-conv kernel (csrc/conv_v2_up.hip)
V2H_CONFIG = 8       # ... of the pre-split kernel's half-height (4 x 32) patch instantiation
V2Q_CONFIG = 9       # ... of its quarter-height (2 x 32) patch instantiation
V2RGB_CONFIG = 13    # ... of the 8-row instantiation that carries the 1x1 head of the forward epilogue (eg3d_conv_v2_params::rgb_out)
V2_UP2 = os.environ.get('EG3D_V2_UP2', '1') != '0'
UP2_MIN_TILES = 256      # workgroups (512 threads, 115 KB of LDS: one per CU) below which the layer stays on the loader-split kernel
UP2_MIN_CK = 32      # 32: SR block 0 conv0 (32 -> 256 channels, 128^2 -> 256^2) too: +0.2 % on the step (A/B twice)


def conv_up2_plan(Ck, Nc, Hi, Wi, N=1):
    """(ksplit, ragged) for the fused-parity transposed-conv kernel on an Hi x Wi -> (2 Hi + 1) x (2 Wi + 1) layer, or None when the layer
    stays on the loader-split kernel.  Measured on MI355X (N = 1): 256^2 x 256 -> 513^2 x 128 172 -> 115 us, 128^2 x 256 -> 257^2 x 128
    64 -> 50 us; the 64^2 / 32^2 layers (64 / 32 workgroups) gain nothing, with or without split-K (fp32 atomics on an output four
    times the size of the input cost what they save), and would pay for the operand split pass on top.
    ragged: the launch covers the full (Hi + 1) x (Wi + 1) cell grid -- chosen when that still fits one round of workgroups (one per
    CU); otherwise the main Hi x Wi grid (perfectly tiled) + the last output row / column as border classes of the loader-split kernel."""
    if not (USE_V2 and V2_UP2) or CONV_MODE != 'auto' or Ck % 16 or Nc % 64 or Ck < UP2_MIN_CK or Hi < 8 or Wi < 32:
        return None
    # 8 x 32-cell patches (eight waves, one workgroup per CU) when they fill the chip; else 4 x 32-cell patches (conv_v2_up2r_kernel: four waves, tap-row weight
    # ring, two workgroups per CU) when THOSE do -- the backbone's 128^2 -> 256^2 layer at one image (128 -> 256 workgroups)
    tiles8 = N * -(-Hi // 8) * -(-Wi // 32) * (Nc // 64)
    tiles4 = N * -(-Hi // 4) * -(-Wi // 32) * (Nc // 64)
    if tiles8 >= UP2_MIN_TILES:
        rows = 8
    elif UP2_ROWS4 and tiles4 >= UP2_MIN_TILES:
        rows = 4
    else:
        return None
    # ragged: the full (Hi + 1) x (Wi + 1) cell grid in one launch when it still fits one round of workgroups (8 rows: one per CU; 4 rows: two per CU)
    ragged = N * -(-(Hi + 1) // rows) * -(-(Wi + 1) // 32) * (Nc // 64) <= (512 if rows == 4 else 256)
    return 1, ragged, rows


# Round 6, A/B in one session (EG3D_UP2_ROWS4=0 | 1 for ALL up2 launches): the SR layers do not care (256^2 x 256 -> 128: 116.9 vs 117.0 us -- 0.99 PFLOP/s executed
# either way, the kernel is at the matrix pipe's sustained rate, not waiting for its epilogue; 128^2 x 32 -> 256: 23.3 vs 27.4 us), the backbone's b256 conv0 leaves the
# loader-split kernel: 64.7 -> 7.9 (operand split) + 50.3 us.  Hence: 4-row patches only where the 8-row grid does not fill the chip.
UP2_ROWS4 = True


def up2_border_classes(Hi, Wi, kh=3, kw=3):
    """The last output row (y = 2 Hi) and column (x = 2 Wi) of the stride-2 transposed 3x3 conv as four tap classes of
    eg3d_conv2d_igemm_f32 (out_stride 2, in_stride 1): cells a = Hi resp. b = Wi, which the 8 x 32-patch grid of conv_up2 leaves out."""
    assert kh == 3 and kw == 3
    wt = lambda ky, kx: ky * 3 + kx                                       # noqa: E731
    return [_mk_class(1, Wi + 1, 2 * Hi, 0, [(Hi - 1, 0, wt(2, 0)), (Hi - 1, -1, wt(2, 2))]),
            _mk_class(1, Wi, 2 * Hi, 1, [(Hi - 1, 0, wt(2, 1))]),
            _mk_class(Hi, 1, 0, 2 * Wi, [(0, Wi - 1, wt(0, 2)), (-1, Wi - 1, wt(2, 2))]),
            _mk_class(Hi, 1, 1, 2 * Wi, [(0, Wi - 1, wt(1, 2))])]


def conv_up2(a: SplitImage, w: SplitImage, out, Hc=None, Wc=None, epi=L.EPI_STORE, ksplit=1, products=3, algo_flops=None, patch_rows=8):
    """Launch eg3d_conv2d_up2: out [N,Co,2Hi+1,2Wi+1] channels_last (+)= transposed 3x3 stride-2 conv of the split image `a` with the
    forward weight image `w`, for the cells a < Hc, b < Wc (default: the full (Hi + 1) x (Wi + 1) grid)."""
    assert is_cl(out)
    p = L.ConvUp2Params()
    n, ck, hi, wi = a.shape
    nc, _, wtaps = w.shape
    assert wtaps == 9
    _, co, ho, wo = out.shape
    p.a, p.w, p.a_scale, p.w_scale, p.out = a.data.data_ptr(), w.data.data_ptr(), a.scale.data_ptr(), w.scale.data_ptr(), out.data_ptr()
    p.N, p.Hi, p.Wi, p.Ck, p.Nc = n, hi, wi, ck, nc
    p.Hc, p.Wc = (hi + 1 if Hc is None else Hc), (wi + 1 if Wc is None else Wc)
    p.Ho, p.Wo, p.ldo = ho, wo, co
    for t in range(9):
        p.wtap[t] = t
    p.epi, p.products, p.ksplit, p.patch_rows = epi, int(products), int(ksplit), int(patch_rows)
    prof = PROFILER
    if prof is not None and prof.only_config is not None and prof.only_config != UP2_CONFIG:
        prof = None
    if prof is not None:
        if algo_flops is None:
            algo_flops = 2.0 * ck * nc * n * hi * wi * 9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().eg3d_conv2d_up2(C.byref(p), L.stream_ptr()), 'conv2d_up2')
    if prof is not None:
        e1.record()
        prof.records.append(((UP2_CONFIG, PRECISIONS['f16x3']), float(algo_flops), e0, e1))
        if prof.meta is not None:
            prof.meta.append(dict(N=n, Hi=hi, Wi=wi, Ck=ck, Nc=nc, Ho=ho, Wo=wo, taps=[4, 2, 2, 1], epi=epi, ksplit=int(ksplit), in_stride=1, out_stride=2,
                                  prec=3, v2=True, up2=True))
    return out


def conv_wgrad(x, g, Ck, Nc, dwp, classes, in_stride=1, out_stride=1, in_scale=None, psplit=0, precision='f32', g_amax=None, g_amax_mul=1.0):
    """dwp[Nc, taps*Ck] += grad of the packed weights (dwp pre-zeroed).  precision 'f32' | 'f16x3' (g_amax: device scalar max|g|
    for the range normalisation of the gradient operand, see include/eg3d_hip.h)."""
    assert is_cl(x) and is_cl(g)
    if len(classes) > 4:        # the launch accumulates into dwp: four classes at a time
        for i in range(0, len(classes), 4):
            conv_wgrad(x, g, Ck, Nc, dwp, classes[i:i + 4], in_stride, out_stride, in_scale, psplit, precision, g_amax, g_amax_mul)
        return dwp
    p = L.WgradParams()
    n, cx, hi, wi = x.shape
    _, cg, ho, wo = g.shape
    p.x, p.g, p.dw = x.data_ptr(), g.data_ptr(), dwp.data_ptr()
    p.N, p.Hi, p.Wi, p.Ck, p.ldx = n, hi, wi, Ck, cx
    p.Nc, p.w_row = Nc, dwp.stride(0)
    p.Ho, p.Wo, p.ldg = ho, wo, cg
    p.in_stride, p.out_stride = in_stride, out_stride
    p.ncls = len(classes)
    for i, c in enumerate(classes):
        p.cls[i] = c
    p.in_scale = in_scale.data_ptr() if in_scale is not None else None
    p.psplit = psplit
    p.precision = PRECISIONS[precision]
    p.g_amax = g_amax.data_ptr() if g_amax is not None else None
    p.g_amax_mul = float(g_amax_mul)
    if DEFERRED_CONV_WGRADS is not None and precision in ('f16x3', 'f16x1'):
        # queued (deferred_weight_grads): launched together with the other layers' by flush_weight_grads; the operands stay referenced until then
        DEFERRED_CONV_WGRADS.append((p, (x, g, dwp, in_scale, g_amax)))
        return dwp
    L.check(L.lib().eg3d_conv2d_wgrad_f32(C.byref(p), L.stream_ptr()), 'conv2d_wgrad_f32')
    return dwp


WGRAD_V2 = os.environ.get('EG3D_WGRAD_V2', '1') != '0'       # weight gradients of split-image layers on csrc/conv_wgrad_v2.hip


def conv_wgrad_v2_ok(gimg, ximg, classes):
    """Both operands as SplitImages of equal geometry, channel counts multiples of 64, one stride-1 class of 9 (3x3) or 1 taps."""
    if not (WGRAD_V2 and USE_V2) or gimg is None or ximg is None or len(classes) != 1 or classes[0].ntaps not in (9, 1):
        return False
    n, co, h, w = gimg.shape
    n2, ci, h2, w2 = ximg.shape
    c = classes[0]
    return (n, h, w) == (n2, h2, w2) and co % 64 == 0 and ci % 64 == 0 and (c.Ha, c.Wa) == (h, w) and c.out_py == 0 and c.out_px == 0


WGRAD_SLABS = os.environ.get('EG3D_WGRAD_SLABS', '0') != '0'   # partial weight-gradient tiles stored to slabs and summed in order (no atomics)


def _wgrad_v2_params(gimg, ximg, classes, products, row_groups):
    p = L.WgradV2Params()
    n, co, h, w = gimg.shape
    _, ci, _, _ = ximg.shape
    c = classes[0]
    p.g, p.x, p.g_scale, p.x_scale = gimg.data.data_ptr(), ximg.data.data_ptr(), gimg.scale.data_ptr(), ximg.scale.data_ptr()
    p.N, p.H, p.W, p.Co, p.Ci, p.w_row = n, h, w, co, ci, c.ntaps * ci
    p.ntaps = c.ntaps
    for t in range(c.ntaps):
        p.dy[t], p.dx[t], p.wtap[t] = c.dy[t], c.dx[t], c.wtap[t]
    p.products, p.row_groups = int(products), int(row_groups)
    return p


def conv_wgrad_v2(gimg: SplitImage, ximg: SplitImage, dwp, classes, products=3, row_groups=0):
    """dwp[Co, taps*Ci] += weight gradient from the split images of dz (gimg) and of the modulated input (ximg) (eg3d_conv2d_wgrad_v2)."""
    p = _wgrad_v2_params(gimg, ximg, classes, products, row_groups)
    p.dw, p.w_row, p.slabs = dwp.data_ptr(), dwp.stride(0), 0
    L.check(L.lib().eg3d_conv2d_wgrad_v2(C.byref(p), L.stream_ptr()), 'conv2d_wgrad_v2')
    return dwp


WGRAD_V2_UP = os.environ.get('EG3D_WGRAD_V2_UP', '1') != '0'
WGRAD_V2_UP_MIN_CELLS = 1024      # input cells (32^2) from which an up layer's weight gradient takes the parity-split kernel


def conv_wgrad_v2_up_ok(Ci, Co, Hi, Wi, N=1):
    """An up-sampling layer's weight gradient on conv_wgrad_v2_up_kernel (parity-split G image x split X image)."""
    return bool(WGRAD_V2 and WGRAD_V2_UP and USE_V2 and CONV_MODE == 'auto' and Ci % 64 == 0 and Co % 64 == 0 and N * Hi * Wi >= WGRAD_V2_UP_MIN_CELLS)


def conv_wgrad_v2_up(gimg: SplitImage, ximg: SplitImage, dwp, wtaps, products=3, row_groups=0):
    """dwp[Co, 9*Ci] += weight gradient of a stride-2 3x3 transposed conv from the parity-split image of its gradient operand (fir44_adjoint_split) and
    the split image of its modulated input; wtaps[3 ky + kx] = weight tap (eg3d_conv2d_wgrad_v2_up)."""
    p = L.WgradV2Params()
    n, ci, h, w = ximg.shape
    co = gimg.shape[1]
    p.g, p.x, p.g_scale, p.x_scale = gimg.data.data_ptr(), ximg.data.data_ptr(), gimg.scale.data_ptr(), ximg.scale.data_ptr()
    p.N, p.H, p.W, p.Co, p.Ci = n, h, w, co, ci
    p.ntaps = 9
    for t in range(9):
        p.dy[t], p.dx[t], p.wtap[t] = 0, 0, int(wtaps[t])
    p.products, p.row_groups, p.slabs = int(products), int(row_groups), 0
    p.dw, p.w_row = dwp.data_ptr(), dwp.stride(0)
    L.check(L.lib().eg3d_conv2d_wgrad_v2_up(C.byref(p), L.stream_ptr()), 'conv2d_wgrad_v2_up')
    return dwp


def conv_wgrad_v2_slabs(gimg: SplitImage, ximg: SplitImage, classes, products=3, row_groups=0):
    """The same without atomics: returns slabs [nslab, Co, taps*Ci] of partial gradients whose sum IN SLAB ORDER is the gradient
    (weight_grad_finish(slabs, ...) sums them; torch.sum(0) for tests)."""
    p = _wgrad_v2_params(gimg, ximg, classes, products, row_groups)
    ns = int(L.lib().eg3d_conv2d_wgrad_v2_slabs(C.byref(p)))
    if ns < 1:
        raise L.Eg3dHipError(f'conv2d_wgrad_v2_slabs: {ns}')
    slabs = torch.empty((ns, p.Co, p.w_row), dtype=torch.float32, device=gimg.data.device)
    p.dw, p.slabs = slabs.data_ptr(), 1
    L.check(L.lib().eg3d_conv2d_wgrad_v2(C.byref(p), L.stream_ptr()), 'conv2d_wgrad_v2')
    return slabs


# ------------------------------------------------------------------------------------------------- epilogues
def epilogue_fwd(z, out, fir=None, pad0=0, fir_gain=1.0, d=None, noise=None, noise_nstride=0, noise_strength=None, bias=None,
                 act='linear', alpha=0.0, gain=1.0, clamp=-1.0, out_amax=None):
    assert is_cl(z) and is_cl(out)
    n, c, h, w = out.shape
    _, _, hz, wz = z.shape
    fh, fw = (fir.shape if fir is not None else (0, 0))
    L.check(L.lib().eg3d_modconv_epilogue_fwd(L.ptr(z), L.ptr(out), n, h, w, c, hz, wz, L.ptr(fir), fh, fw, pad0, float(fir_gain),
                                              L.ptr(d), L.ptr(noise), noise_nstride, L.ptr(noise_strength), L.ptr(bias),
                                              L.ACT_IDS[act], float(alpha), float(gain), float(clamp), L.ptr(out_amax), L.stream_ptr()),
            'modconv_epilogue_fwd')
    return out


UPCONV_EPI = os.environ.get('EG3D_UPCONV_EPI', '1') != '0'       # LDS-staged separable FIR epilogue for the up layers (eg3d_upconv_epilogue_fwd)
UPCONV_EPI_SPLIT = os.environ.get('EG3D_UPCONV_EPI_SPLIT', '1') != '0'   # ... that also writes the consumer's split operand image (clamp-bounded layers)
_K4 = (C.c_float * 4)(0.125, 0.375, 0.375, 0.125)


def upconv_epilogue_fwd(z, out, pad0=1, fir_gain=4.0, d=None, noise=None, noise_nstride=0, noise_strength=None, bias=None, act='linear', alpha=0.0, gain=1.0,
                        clamp=-1.0, out_amax=None, split_in_scale=None):
    """eg3d_upconv_epilogue_fwd with the [1,3,3,1] / 8 taps.  split_in_scale (the consumer layer's styles [N,C]; needs clamp >= 0): also returns
    that layer's SplitImage of `out`."""
    assert is_cl(z) and is_cl(out)
    n, c, h, w = out.shape
    _, _, hz, wz = z.shape
    img = scale = None
    if split_in_scale is not None:
        img = torch.empty((n * h * w * c * 2,), dtype=torch.float16, device=out.device)
        scale = torch.empty((1,), dtype=torch.float32, device=out.device)
    L.check(L.lib().eg3d_upconv_epilogue_fwd(L.ptr(z), L.ptr(out), n, h, w, c, hz, wz, _K4, pad0, float(fir_gain), L.ptr(d), L.ptr(noise), noise_nstride,
                                             L.ptr(noise_strength), L.ptr(bias), L.ACT_IDS[act], float(alpha), float(gain), float(clamp), L.ptr(out_amax),
                                             L.ptr(split_in_scale), L.ptr(img), L.ptr(scale), L.stream_ptr()), 'upconv_epilogue_fwd')
    return SplitImage(img, scale, (n, c, h, w)) if img is not None else None


def epilogue_bwd(dout, out, dz, d=None, noise=None, noise_nstride=0, noise_strength=None, bias=None, act='linear', alpha=0.0, gain=1.0,
                 clamp=-1.0, dbias=None, dd=None, dnoise=None, dnoise_nstride=0, dstrength=None, dz_amax=None):
    assert is_cl(dout) and is_cl(out) and is_cl(dz)
    n, c, h, w = out.shape
    L.check(L.lib().eg3d_modconv_epilogue_bwd(L.ptr(dout), L.ptr(out), L.ptr(dz), n, h, w, c, L.ptr(d), L.ptr(noise), noise_nstride,
                                              L.ptr(noise_strength), L.ptr(bias), L.ACT_IDS[act], float(alpha), float(gain),
                                              float(clamp), L.ptr(dbias), L.ptr(dd), L.ptr(dnoise), dnoise_nstride, L.ptr(dstrength),
                                              L.ptr(dz_amax), L.stream_ptr()), 'modconv_epilogue_bwd')
    return dz


def dgrad_finish(z, x, s, dx, ds=None, addend=None):
    n, c, h, w = z.shape
    L.check(L.lib().eg3d_dgrad_finish(L.ptr(z), L.ptr(x), L.ptr(s), L.ptr(addend), L.ptr(dx), L.ptr(ds), n, h, w, c, L.stream_ptr()), 'dgrad_finish')
    return dx


def dgrad_finish_act(z, x, s, dz, act_bwd, ds=None, addend=None, dz_amax=None):
    """eg3d_dgrad_finish_act: split-K finish + the activation backward of the layer that produced x; dz receives THAT layer's dz."""
    n, c, h, w = z.shape
    ab = L.ActBwd()
    act_bwd.fill(ab)
    L.check(L.lib().eg3d_dgrad_finish_act(L.ptr(z), L.ptr(x), L.ptr(s), L.ptr(addend), L.ptr(dz), L.ptr(ds), n, h, w, c, C.byref(ab), L.ptr(dz_amax),
                                          L.stream_ptr()), 'dgrad_finish_act')
    return dz


def torgb_dgrad_act(dy4, wa4, x, s, dz, act_bwd, ds=None, addend=None, dz_amax=None):
    """eg3d_torgb_dgrad_act: data gradient of a 1x1 layer with four (padded) outputs + the activation backward of the layer that produced x,
    one element-wise pass.  dy4 [N,4,H,W] channels_last, wa4 [C,4] contiguous; dz receives the producing layer's dz."""
    n, c, h, w = x.shape
    assert is_cl(dy4) and dy4.shape[1] == 4 and tuple(wa4.shape) == (c, 4) and wa4.is_contiguous()
    ab = L.ActBwd()
    act_bwd.fill(ab)
    L.check(L.lib().eg3d_torgb_dgrad_act(L.ptr(dy4), L.ptr(wa4), L.ptr(x), L.ptr(s), L.ptr(addend), L.ptr(dz), L.ptr(ds), n, h, w, c, C.byref(ab), L.ptr(dz_amax),
                                         L.stream_ptr()), 'torgb_dgrad_act')
    return dz


def torgb_dgrad_act_split(dy4, wa4, x, s, act_bwd, dy_amax, ds=None, addend=None, addend_amax=None, dz=None):
    """eg3d_torgb_dgrad_act_split: torgb_dgrad_act whose dz leaves as the SplitImage the consuming data gradient reads (no split pass, and no
    fp32 dz unless `dz` is given).  dy_amax / addend_amax: device scalars max|dy4| / max|addend|."""
    n, c, h, w = x.shape
    assert is_cl(dy4) and dy4.shape[1] == 4 and tuple(wa4.shape) == (c, 4) and wa4.is_contiguous() and c % 8 == 0
    ab = L.ActBwd()
    act_bwd.fill(ab)
    img = torch.empty((n * h * w * c * 2,), dtype=torch.float16, device=x.device)
    scale = torch.empty((1,), dtype=torch.float32, device=x.device)
    L.check(L.lib().eg3d_torgb_dgrad_act_split(L.ptr(dy4), L.ptr(wa4), L.ptr(x), L.ptr(s), L.ptr(addend), L.ptr(dz), L.ptr(ds), n, h, w, c, C.byref(ab),
                                               L.ptr(dy_amax), L.ptr(addend_amax), L.ptr(img), L.ptr(scale), L.stream_ptr()), 'torgb_dgrad_act_split')
    return SplitImage(img, scale, (n, c, h, w))


def rows_gram(a, b, out_scale=1.0, colsum_scale=1.0):
    """(out_scale * a^T b [Ka,Kb], colsum_scale * column sums of a [Ka]) for row matrices a [S,Ka], b [S,Kb] with Ka, Kb <= 64 (eg3d_rows_gram_scaled)."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.is_contiguous() and b.is_contiguous()
    z = zeros((a.shape[1] * b.shape[1] + a.shape[1],), a.device)
    out, cs = z[:a.shape[1] * b.shape[1]].view(a.shape[1], b.shape[1]), z[a.shape[1] * b.shape[1]:]
    L.check(L.lib().eg3d_rows_gram_scaled(L.ptr(a), L.ptr(b), a.shape[0], a.shape[1], b.shape[1], L.ptr(out), L.ptr(cs), float(out_scale), float(colsum_scale),
                                          L.stream_ptr()), 'rows_gram')
    return out, cs


def style_affine(ws, layers, outs=None, douts=None, dws=None, demod=None, backward=False, dweights=None, dbiases=None):
    """eg3d_style_affine_fwd / _bwd.  ws: [N,L,D] contiguous fp32; layers: sequence of (weight [C,D], bias [C] | None, wrow, wgain,
    bgain, post); outs: per-layer [N,C] styles (written by fwd, read by bwd); demod: per layer None or (wsq [Co,C], d [N,Co],
    dd [N,Co] | None, dout_extra [N,C] | None) -- the demodulation coefficients of the layer's conv and their backward;
    dweights / dbiases: per layer None or the tensor that receives the gradient of that layer's affine weight / bias (backward)."""
    assert ws.is_contiguous() and ws.dtype == torch.float32 and len(layers) <= L.STYLE_BANK_MAX
    b = L.StyleBank()
    b.ws, b.N, b.L, b.D, b.nlayers = ws.data_ptr(), ws.shape[0], ws.shape[1], ws.shape[2], len(layers)
    b.dws = dws.data_ptr() if dws is not None else None
    for i, (w, bias, wrow, wgain, bgain, post) in enumerate(layers):
        ly = b.layers[i]
        assert w.is_contiguous() and w.dtype == torch.float32 and w.shape[1] == b.D
        ly.weight, ly.bias = w.data_ptr(), (bias.data_ptr() if bias is not None else None)
        ly.C, ly.wrow, ly.wgain, ly.bgain, ly.post = w.shape[0], int(wrow), float(wgain), float(bgain), float(post)
        ly.out = outs[i].data_ptr() if outs is not None else None
        ly.dout = douts[i].data_ptr() if (douts is not None and douts[i] is not None) else None
        if dweights is not None and dweights[i] is not None:
            assert dweights[i].is_contiguous() and dweights[i].shape == w.shape
            ly.dweight = dweights[i].data_ptr()
        if dbiases is not None and dbiases[i] is not None:
            ly.dbias = dbiases[i].data_ptr()
        dm = demod[i] if demod is not None else None
        if dm is not None:
            wsq, d, dd, extra = dm
            assert wsq.is_contiguous() and wsq.shape[1] == w.shape[0]
            ly.Co, ly.wsq, ly.d = wsq.shape[0], wsq.data_ptr(), d.data_ptr()
            ly.dd = dd.data_ptr() if dd is not None else None
            ly.dout_extra = extra.data_ptr() if extra is not None else None
    fn = L.lib().eg3d_style_affine_bwd if backward else L.lib().eg3d_style_affine_fwd
    L.check(fn(C.byref(b), L.stream_ptr()), 'style_affine')


def pack_conv_weight(w):
    """[O,I,kh,kw] fp32 contiguous -> (wf [O, taps*I], wa [I, taps*O], wsq [O,I]) in one launch (eg3d_pack_conv_weight)."""
    L.require_cuda(w)
    w = w.detach().contiguous().float()
    o, i, kh, kw = w.shape
    wf = torch.empty((o, kh * kw * i), device=w.device)
    wa = torch.empty((i, kh * kw * o), device=w.device)
    wsq = torch.empty((o, i), device=w.device)
    L.check(L.lib().eg3d_pack_conv_weight(w.data_ptr(), wf.data_ptr(), wa.data_ptr(), wsq.data_ptr(), o, i, kh * kw, L.stream_ptr()), 'pack_conv_weight')
    return wf, wa, wsq


def pack_conv_weight_padded(w, wf_p, wa_p, o_pad):
    """eg3d_pack_conv_weight_padded into caller-owned, zero-initialised buffers wf_p [o_pad, taps*I], wa_p [I, taps*o_pad]."""
    L.require_cuda(w, wf_p, wa_p)
    w = w.detach().contiguous().float()
    o, i, kh, kw = w.shape
    assert tuple(wf_p.shape) == (o_pad, kh * kw * i) and tuple(wa_p.shape) == (i, kh * kw * o_pad) and wf_p.is_contiguous() and wa_p.is_contiguous()
    L.check(L.lib().eg3d_pack_conv_weight_padded(w.data_ptr(), wf_p.data_ptr(), wa_p.data_ptr(), None, o, i, kh * kw, o_pad, L.stream_ptr()),
            'pack_conv_weight_padded')


def split_weight_pieces(wp):
    """The two-piece fp16 image of a packed weight matrix for conv_igemm(..., w_pieces=): element group [4j..4j+3] -> 4 high + 4 low fp16
    pieces in the same 16 bytes (eg3d_split_weight_pieces).  Returned as a float32-typed tensor of wp's shape (it holds fp16 bit patterns)."""
    L.require_cuda(wp)
    assert wp.is_contiguous() and wp.dtype == torch.float32 and wp.numel() % 4 == 0
    img = torch.empty_like(wp)
    L.check(L.lib().eg3d_split_weight_pieces(wp.data_ptr(), img.data_ptr(), wp.numel(), L.stream_ptr()), 'split_weight_pieces')
    return img


def pack_conv_weights_batched(items):
    """items: list of (w [O,I,kh,kw], wf, wa | None, wsq | None, o_pad[, oscale [O] | None]) with caller-allocated outputs (eg3d_pack_conv_weights_batched);
    one launch per L.PACK_BATCH_MAX layers."""
    for a in range(0, len(items), L.PACK_BATCH_MAX):
        chunk = items[a:a + L.PACK_BATCH_MAX]
        arr = (L.PackItem * len(chunk))()
        for q, item in zip(arr, chunk):
            w, wf, wa, wsq, o_pad = item[:5]
            oscale = item[5] if len(item) > 5 else None          # [O]: folded into both images (eg3d_pack_item::oscale)
            L.require_cuda(w, wf)
            assert w.is_contiguous() and w.dtype == torch.float32
            o, i, kh, kw = w.shape
            q.w, q.wf, q.wa, q.wsq = w.data_ptr(), wf.data_ptr(), L.ptr(wa), L.ptr(wsq)
            q.O, q.I, q.T, q.O_pad = o, i, kh * kw, int(o_pad)
            q.oscale = L.ptr(oscale)
        L.check(L.lib().eg3d_pack_conv_weights_batched(arr, len(chunk), L.stream_ptr()), 'pack_conv_weights_batched')


DEFERRED_WGRADS = None         # a list while `deferred_weight_grads()` is active: weight_grad_finish() queues its arguments there and returns None
DEFERRED_CONV_WGRADS = None    # ... and conv_wgrad() its launch parameters (eg3d_conv2d_wgrad_batched: the small layers fill the chip only together)
BATCH_CONV_WGRADS = os.environ.get('EG3D_BATCH_CONV_WGRADS', '1') != '0'


@contextlib.contextmanager
def deferred_weight_grads():
    """Pivotal tuning: the conv layers' backward passes queue their packed weight-gradient images instead of launching one
    eg3d_weight_grad_finish each (17 launches of 5 - 13 us); `flush_weight_grads(queue)` turns all of them into the parameters' `.grad` with one
    launch (eg3d_weight_grad_finish_batched).  Inside the context the layers return NO gradient for their weights through autograd --
    only a caller that flushes before its optimiser step may use it (inversion.PivotalTuner does)."""
    global DEFERRED_WGRADS, DEFERRED_CONV_WGRADS
    prev, DEFERRED_WGRADS = DEFERRED_WGRADS, []
    prev_c, DEFERRED_CONV_WGRADS = DEFERRED_CONV_WGRADS, ([] if BATCH_CONV_WGRADS else None)
    DEFERRED_WGRADS.append(DEFERRED_CONV_WGRADS)          # (slot 0 of the queue: the conv launches, which the finishing pass depends on)
    try:
        yield DEFERRED_WGRADS
    finally:
        DEFERRED_WGRADS, DEFERRED_CONV_WGRADS = prev, prev_c


_WGRAD_BUFS = {}               # id(weight) -> (weakref(weight), persistent gradient buffer): the same storage every step (graph replay)


def flush_weight_grads(queue):
    """One launch for everything `deferred_weight_grads()` queued; sets / accumulates `weight.grad`."""
    if not queue:
        return
    convs, queue = queue[0], queue[1:]
    if convs:              # first the queued weight-gradient GEMMs themselves, five layers per launch
        arr = (L.WgradParams * len(convs))(*[p for p, _ in convs])
        L.check(L.lib().eg3d_conv2d_wgrad_batched(arr, len(convs), L.stream_ptr()), 'conv2d_wgrad_batched')
        keep_for_capture(*[t for _, ts in convs for t in ts if t is not None])
    for a in range(0, len(queue), L.WGF_BATCH_MAX):
        chunk = queue[a:a + L.WGF_BATCH_MAX]
        arr = (L.WgfItem * len(chunk))()
        outs = []
        for q, (dwp, weight, styles, d, dd) in zip(arr, chunk):
            w = weight.detach()
            assert w.is_contiguous() and w.dtype == torch.float32
            o, i, kh, kw = w.shape
            ent = _WGRAD_BUFS.get(id(weight))
            if ent is None or ent[0]() is not weight or ent[1].shape != w.shape:
                if len(_WGRAD_BUFS) > 512:
                    _WGRAD_BUFS.clear()
                ent = (weakref.ref(weight), torch.empty_like(w))
                _WGRAD_BUFS[id(weight)] = ent
            dw = ent[1]
            q.g, q.w, q.s, q.d, q.dd, q.dw = dwp.data_ptr(), w.data_ptr(), L.ptr(styles), L.ptr(d), L.ptr(dd), dw.data_ptr()
            q.N, q.O, q.I, q.T = (styles.shape[0] if styles is not None else 1), o, i, kh * kw
            if dwp.dim() == 3:
                assert dwp.is_contiguous() and dwp.shape[1:] == (o, kh * kw * i)
                q.nslab, q.slab_stride = dwp.shape[0], dwp.stride(0)
            else:
                q.nslab, q.slab_stride = 1, 0
            outs.append((weight, dw))
        L.check(L.lib().eg3d_weight_grad_finish_batched(arr, len(chunk), L.stream_ptr()), 'weight_grad_finish_batched')
        keep_for_capture(*[t for item in chunk for t in item if torch.is_tensor(t)])
        for weight, dw in outs:
            if weight.grad is None:
                weight.grad = dw
            elif weight.grad.data_ptr() != dw.data_ptr():
                weight.grad.add_(dw)
            # (else: the optimiser kept last step's buffer as .grad -- it now holds this step's gradient)


def weight_grad_finish(dwp, weight, styles, d, dd):
    """[O,I,kh,kw] gradient of a demodulated modulated conv weight from the packed weight-gradient image dwp [O, taps*I] and the
    demodulation path (eg3d_weight_grad_finish); dd None: no demodulation term."""
    L.require_cuda(dwp, weight)
    if DEFERRED_WGRADS is not None and weight.is_contiguous() and weight.dtype == torch.float32:
        DEFERRED_WGRADS.append((dwp, weight, styles, d, dd))
        return None
    w = weight.detach().contiguous().float()
    o, i, kh, kw = w.shape
    dw = torch.empty_like(w)
    n = styles.shape[0] if styles is not None else 1
    if dwp.dim() == 3:          # slabs [nslab, O, taps*I] (conv_wgrad_v2_slabs): summed in slab order by the same pass
        assert dwp.is_contiguous() and dwp.shape[1:] == (o, kh * kw * i)
        L.check(L.lib().eg3d_weight_grad_finish_slabs(dwp.data_ptr(), dwp.shape[0], dwp.stride(0), w.data_ptr(), L.ptr(styles), L.ptr(d), L.ptr(dd),
                                                      dw.data_ptr(), n, o, i, kh * kw, L.stream_ptr()), 'weight_grad_finish_slabs')
        return dw
    L.check(L.lib().eg3d_weight_grad_finish(dwp.data_ptr(), w.data_ptr(), L.ptr(styles), L.ptr(d), L.ptr(dd), dw.data_ptr(), n, o, i, kh * kw,
                                            L.stream_ptr()), 'weight_grad_finish')
    return dw


def weight_sqsum(wp, Co, ntaps, Ck):
    wsq = torch.empty((Co, Ck), dtype=torch.float32, device=wp.device)
    L.check(L.lib().eg3d_weight_sqsum(L.ptr(wp), L.ptr(wsq), Co, ntaps, Ck, L.stream_ptr()), 'weight_sqsum')
    return wsq


def demod_fwd(s, wsq):
    n, ck = s.shape
    co = wsq.shape[0]
    d = torch.empty((n, co), dtype=torch.float32, device=s.device)
    L.check(L.lib().eg3d_demod_fwd(L.ptr(s), L.ptr(wsq), L.ptr(d), n, co, ck, L.stream_ptr()), 'demod_fwd')
    return d


def demod_bwd(s, wsq, d, dd, ds=None, dwsq=None):
    n, ck = s.shape
    co = wsq.shape[0]
    L.check(L.lib().eg3d_demod_bwd(L.ptr(s), L.ptr(wsq), L.ptr(d), L.ptr(dd), L.ptr(ds), L.ptr(dwsq), n, co, ck, L.stream_ptr()),
            'demod_bwd')


# ------------------------------------------------------------------------------------------------- renderer
def ray_gen_fwd(c2w, K, res):
    n = c2w.shape[0]
    o = torch.empty((n, res * res, 3), dtype=torch.float32, device=c2w.device)
    d = torch.empty_like(o)
    L.check(L.lib().eg3d_ray_gen_fwd(L.ptr(c2w), L.ptr(K), L.ptr(o), L.ptr(d), n, res, L.stream_ptr()), 'ray_gen_fwd')
    return o, d


def ray_gen_bwd(c2w, K, g_o, g_d, res, want_K=True):
    n = c2w.shape[0]
    d_c2w = torch.empty((n, 4, 4), dtype=torch.float32, device=c2w.device)
    d_K = torch.empty((n, 3, 3), dtype=torch.float32, device=c2w.device) if want_K else None
    L.check(L.lib().eg3d_ray_gen_bwd(L.ptr(c2w), L.ptr(K), L.ptr(g_o), L.ptr(g_d), L.ptr(d_c2w), L.ptr(d_K), n, res, L.stream_ptr()),
            'ray_gen_bwd')
    return d_c2w, d_K


SCATTER_F16 = True      # tri-plane gradient accumulation with three fp16 products per fp32 product (scatter_accum16h): 128 -> 112 us, C2 +0.4 % (round 6, three alternating pairs); False = exact fp32 products (scatter_accum16p)


def make_render_params(planes, origins, dirs, u1, u2, opts, w0, b0, w1t, b1, rgb, depth, wsum, minmax, fine, ray_limits=None, save=None, ray_tile_width=None, pos_rows=None,
                       feat_rows=None, dbg=None):
    """planes: channels_last [N, 3*C, Hp, Wp]; decoder weights with gains folded (w1t transposed [H, 1+Cout])."""
    assert is_cl(planes)
    p = L.RenderParams()
    n, c3, hp, wp = planes.shape
    p.planes, p.N, p.Hp, p.Wp, p.ldp, p.C = planes.data_ptr(), n, hp, wp, c3, c3 // 3
    p.origins, p.dirs, p.R = origins.data_ptr(), dirs.data_ptr(), origins.shape[1]
    p.u1 = u1.data_ptr()
    p.u2 = u2.data_ptr() if u2 is not None else None
    p.Dc, p.Df = int(opts['depth_resolution']), int(opts['depth_resolution_importance'])
    if ray_limits is not None:
        p.ray_limits = ray_limits.data_ptr()
        p.ray_start = p.ray_end = 0.0
    else:
        p.ray_limits = None
        p.ray_start, p.ray_end = float(opts['ray_start']), float(opts['ray_end'])
    p.disparity = int(bool(opts.get('disparity_space_sampling', False)))
    p.box_warp = float(opts['box_warp'])
    p.white_back = int(bool(opts.get('white_back', False)))
    p.w0, p.b0, p.w1, p.b1 = w0.data_ptr(), b0.data_ptr(), w1t.data_ptr(), b1.data_ptr()
    p.Hdim, p.Cout = w0.shape[0], w1t.shape[1] - 1
    p.rgb = rgb.data_ptr() if rgb is not None else None
    p.depth = depth.data_ptr() if depth is not None else None
    p.wsum = wsum.data_ptr() if wsum is not None else None
    p.depth_minmax = minmax.data_ptr() if minmax is not None else None
    p.fine_depths = fine.data_ptr() if fine is not None else None
    if save is not None:
        p.save_sigma, p.save_rgb = save[0].data_ptr(), save[1].data_ptr()
    if ray_tile_width is None:          # RaySampler emits the pixels of a square image in row-major order (ray_sampler.py:43-48)
        side = int(round(p.R ** 0.5))
        ray_tile_width = side if side * side == p.R and side % 32 == 0 else 0
    p.ray_tile_width = int(ray_tile_width)
    p.pos_rows = pos_rows.data_ptr() if pos_rows is not None else None       # workspace [2, N*R, D, 4]: selects the pipelined forward
    p.feat_rows = feat_rows.data_ptr() if feat_rows is not None else None    # [S, 32]: gather as its own pass, rows re-read by the decoder kernels
    if dbg is not None:                  # (int32 [N*R, Df, 3], int32 [N*R, Dc + Df]): the sampler's integer side (include/eg3d_hip.h)
        p.dbg_inds, p.dbg_ranks = dbg[0].data_ptr(), dbg[1].data_ptr()
        p.dbg_cdf = dbg[2].data_ptr() if len(dbg) > 2 and dbg[2] is not None else None
    return p


def render_sizes(p):
    """Element counts of every caller-owned renderer buffer (eg3d_render_query_sizes): the host allocates from the library's own answer."""
    z = L.RenderSizes()
    L.check(L.lib().eg3d_render_query_sizes(C.byref(p), C.byref(z)), 'render_query_sizes')
    return z


def render_fwd(p):
    L.check(L.lib().eg3d_render_fwd(C.byref(p), L.stream_ptr()), 'render_fwd')


def render_finalize(depth, minmax):
    L.check(L.lib().eg3d_render_finalize(L.ptr(depth), L.ptr(minmax), depth.numel(), L.stream_ptr()), 'render_finalize')


def render_bwd(p, d_rgb, d_depth, d_wsum, d_planes, d_origins, d_dirs, dumps=None, gram=None):
    """d_planes (pre-zeroed, channels_last [N,3C,Hp,Wp]) is filled by the tile-binned scatter of the dumped rows."""
    bp = L.RenderBwdParams()
    bp.fwd = p
    bp.depth_out = None
    bp.d_rgb = d_rgb.data_ptr()
    bp.d_depth = d_depth.data_ptr() if d_depth is not None else None
    bp.d_wsum = d_wsum.data_ptr() if d_wsum is not None else None
    rows = pos = None
    D = max(p.Dc, p.Df)
    z = render_sizes(p)
    S = z.S
    dev = d_rgb.device
    ag = torch.empty(z.ag_rows, dtype=torch.float32, device=dev)
    bp.ag_rows = ag.data_ptr()
    if d_origins is not None or d_dirs is not None:
        gc = torch.empty(z.gc_rows, dtype=torch.float32, device=dev)
        bp.gc_rows = gc.data_ptr()
    pos = torch.empty(z.df_pos, dtype=torch.float32, device=dev)        # (x, y, z, depth) per sample row, NaN = absent
    bp.df_pos = pos.data_ptr()
    if d_planes is not None:
        rows = torch.empty(z.df_rows, dtype=torch.float32, device=dev)
        bp.df_rows = rows.data_ptr()
    bp.d_origins = d_origins.data_ptr() if d_origins is not None else None
    bp.d_dirs = d_dirs.data_ptr() if d_dirs is not None else None
    if dumps is not None:
        bp.dump_dpre, bp.dump_h, bp.dump_dout, bp.dump_feat = [t.data_ptr() for t in dumps]
    if gram is not None:        # (w0 [64,32], b0 [64], w1 [33,64], b1 [33] pre-zeroed, scale0, scale1, bias_scale): contracted inside the sample-level kernel
        bp.gram_w0, bp.gram_b0, bp.gram_w1, bp.gram_b1 = [t.data_ptr() for t in gram[:4]]
        bp.gram_scale0, bp.gram_scale1, bp.gram_bias_scale = float(gram[4]), float(gram[5]), float(gram[6])
        if p.feat_rows and bp.dump_feat == p.feat_rows:         # the caller uses the saved feature rows as that operand: nothing to dump
            bp.dump_feat = None
    amax = None
    if d_planes is not None and SCATTER_F16:        # plane-gradient accumulation on the 16-bit matrix cores: needs max|df_rows| (written by the call)
        amax = torch.empty(1, dtype=torch.float32, device=dev)
        bp.df_amax = amax.data_ptr()
    L.check(L.lib().eg3d_render_bwd(C.byref(bp), L.stream_ptr()), 'render_bwd')
    if d_planes is not None:
        nints = L.lib().eg3d_triplane_scatter_workspace_ints(S, p.N, p.Hp, p.Wp)
        ws = torch.empty(nints, dtype=torch.int32, device=d_planes.device)
        rw = math.isqrt(p.R)            # rays are generated row by row over a square image (ray_sampler.py:41-50): a hint for the binning order only
        L.check(L.lib().eg3d_triplane_scatter(L.ptr(rows), L.ptr(pos), S, p.R * 2 * D, L.ptr(d_planes), p.N, p.Hp, p.Wp, p.ldp,
                                              p.box_warp, L.ptr(ws), rw if rw * rw == p.R else 0, 2 * D, L.ptr(amax) if amax is not None else None,
                                              L.stream_ptr()), 'triplane_scatter')


def sample_decode(p, coords, M):
    n = p.N
    rgb = torch.empty((n, M, p.Cout), dtype=torch.float32, device=coords.device)
    sigma = torch.empty((n, M, 1), dtype=torch.float32, device=coords.device)
    L.check(L.lib().eg3d_sample_decode(C.byref(p), L.ptr(coords), M, L.ptr(rgb), L.ptr(sigma), L.stream_ptr()), 'sample_decode')
    return rgb, sigma


# ------------------------------------------------------------------------------------------------- noise buffers
def _buf_arrays(bufs):
    n = len(bufs)
    xs = (C.c_void_p * n)(*[b.data_ptr() for b in bufs])
    res = (C.c_int32 * n)(*[int(b.shape[-1]) for b in bufs])
    return n, xs, res


NOISE_BANK_MAX = 32          # buffers per launch of the noise kernels (include/eg3d_hip.h)


def noise_regularizer(bufs, scale=1.0, want_grad=True, grads=None):
    """Returns (reg [0-d tensor] = scale * regulariser summed over all buffers, grads list or None) for square fp32 noise buffers;
    one launch per 32 buffers (a batch of images brings 17 per image).  `grads`: optional pre-allocated gradient tensors (views allowed)."""
    for b in bufs:
        L.require_cuda(b)
        assert b.dim() == 2 and b.shape[0] == b.shape[1] and b.is_contiguous() and b.dtype == torch.float32
    dev = bufs[0].device
    if want_grad and grads is None:
        grads = [torch.empty_like(b) for b in bufs]
    total = None
    for lo in range(0, len(bufs), NOISE_BANK_MAX):
        part = bufs[lo:lo + NOISE_BANK_MAX]
        n, xs, res = _buf_arrays(part)
        gs = (C.c_void_p * n)(*[g.data_ptr() for g in grads[lo:lo + NOISE_BANK_MAX]]) if want_grad else (C.c_void_p * n)()
        ws = torch.empty(int(L.lib().eg3d_noise_reg_workspace_floats(res, n)), dtype=torch.float32, device=dev)
        reg = torch.empty((), dtype=torch.float32, device=dev)
        L.check(L.lib().eg3d_noise_regularizer(xs, gs, res, n, L.ptr(ws), L.ptr(reg), float(scale), L.stream_ptr()), 'noise_regularizer')
        total = reg if total is None else total + reg
    return total, (grads if want_grad else None)



class PendingEpilogue:
    """A 3 x 3 layer's finishing epilogue that has not run yet: the layer's output tensor is allocated but still unwritten, `z` holds the split-K
    sums.  The toRGB launch that consumes the output next runs it (torgb_small(pre=...)); anything else calls run() first."""
    __slots__ = ('z', 'out', 'd', 'noise', 'noise_nstride', 'noise_strength', 'bias', 'act', 'alpha', 'gain', 'clamp', 'out_amax')

    def __init__(self, z, out, d, out_amax, noise=None, noise_nstride=0, noise_strength=None, bias=None, act='linear', alpha=0.0, gain=1.0, clamp=-1.0):
        self.z, self.out, self.d, self.out_amax = z, out, d, out_amax
        self.noise, self.noise_nstride, self.noise_strength, self.bias = noise, noise_nstride, noise_strength, bias
        self.act, self.alpha, self.gain, self.clamp = act, alpha, gain, clamp

    def run(self):
        epilogue_fwd(self.z, self.out, d=self.d, out_amax=self.out_amax, noise=self.noise, noise_nstride=self.noise_nstride, noise_strength=self.noise_strength,
                     bias=self.bias, act=self.act, alpha=self.alpha, gain=self.gain, clamp=self.clamp)


# conv1's finishing pass inside the toRGB launch of the 4^2 .. 64^2 blocks (eg3d_torgb_small_params::pre_z): five launches fewer per step.  The first
# version was NOT faster (merged launch 17.9 - 18.9 us against 10.0 - 10.9 + 5.5 - 9.1 for the two): its d / bias loads sat under (uniform) null-pointer
# branches inside the unrolled batch, and a load under a branch is a memory round trip of its own -- sixteen in sequence per batch.  Issued with the
# operand loads (dummy address when absent) the merged launch is 14.2 - 15.3 us and the step +0.5 % (220.1 vs 219.0 steps/s, A/B).  EG3D_DEFER_EPILOGUE=0: separate pass.
DEFER_EPILOGUE = os.environ.get('EG3D_DEFER_EPILOGUE', '1') != '0'
# the mirror image in the backward pass: conv0's split-K data gradient goes to the previous block's toRGB node unfinished (fused.PENDING_DGRAD) and that
# node's launch applies the styles and accumulates the style gradient (eg3d_torgb_small_bwd_params::add_scale): the dgrad_finish launch of the 8^2 .. 64^2 blocks is gone
DEFER_DGRAD_FINISH = os.environ.get('EG3D_DEFER_DGRAD_FINISH', '1') != '0'

TORGB_SMALL = os.environ.get('EG3D_TORGB_SMALL', '1') != '0'            # low-latency toRGB launch for small pixel counts (csrc/torgb_small.hip)
TORGB_SMALL_MAX_PIX = 4096
# (the data gradient at 64^2 takes 31 us in the trace against 27 for the implicit GEMM it replaced -- yet the step is 0.2 % faster with it: A/B 209.1 vs 208.8)
TORGB_SMALL_BWD_MAX_PIX = 4096
# the streaming form of the same entry points for the 128^2 / 256^2 blocks (torgb_mid_kernel / torgb_mid_bwd_kernel, csrc/torgb_small.hip): beyond
# TORGB_SMALL_*_MAX_PIX the launch is taken only when the library says it runs in that form (eg3d_torgb_mid_supported); EG3D_TORGB_MID=0: implicit GEMM as before
TORGB_MID = os.environ.get('EG3D_TORGB_MID', '1') != '0'
TORGB_MID_BWD = os.environ.get('EG3D_TORGB_MID_BWD', '1') != '0'


def torgb_small(x, wf, styles, out, bias=None, clamp=-1.0, addend=None, addend_up2_taps=None, pre=None):
    """eg3d_torgb_small_fwd: out = clamp(conv1x1(x * styles, wf) + bias) + addend.  x / out / addend channels_last fp32; wf [Cp, C] packed rows.
    `pre` (a PendingEpilogue): x does not exist yet -- the launch runs the producing layer's finishing epilogue on its split-K sums and WRITES x.
    Returns False (nothing launched) when the geometry is not the kernel's."""
    assert is_cl(x) and is_cl(out)
    n, c, h, w = x.shape
    p = L.TorgbSmallParams(x=x.data_ptr(), w=wf.data_ptr(), s=styles.data_ptr(), bias=bias.data_ptr() if bias is not None else None,
                           addend=addend.data_ptr() if addend is not None else None, out=out.data_ptr(), N=n, H=h, W=w, C=c, Cp=out.shape[1],
                           ldx=c, ldo=out.shape[1], w_row=wf.stride(0), addend_up2=1 if addend_up2_taps is not None else 0, clamp=float(clamp))
    if addend_up2_taps is not None:
        p.addend_taps[:] = [float(t) for t in addend_up2_taps]
    if pre is not None:
        if pre.act not in ('linear', 'lrelu', 'relu') or tuple(pre.z.shape) != tuple(x.shape) or not is_cl(pre.z):
            return False
        p.pre_z, p.pre_d, p.pre_bias = pre.z.data_ptr(), L.ptr(pre.d), L.ptr(pre.bias)
        p.pre_noise, p.pre_strength, p.pre_noise_nstride = L.ptr(pre.noise), L.ptr(pre.noise_strength), int(pre.noise_nstride or 0)
        p.x_amax = L.ptr(pre.out_amax)
        p.pre_slope = {'linear': 1.0, 'lrelu': float(pre.alpha), 'relu': 0.0}[pre.act]
        p.pre_gain, p.pre_clamp = float(pre.gain), float(pre.clamp)
    if not L.lib().eg3d_torgb_small_supported(C.byref(p)):
        return False
    if n * h * w > TORGB_SMALL_MAX_PIX and not (TORGB_MID and L.lib().eg3d_torgb_mid_supported(C.byref(p))):
        return False
    L.check(L.lib().eg3d_torgb_small_fwd(C.byref(p), L.stream_ptr()), 'torgb_small_fwd')
    return True


def torgb_small_bwd(dy, wa, styles, x, dx, ds=None, addend=None, act_bwd=None, out_amax=None, addend_scale=None, addend_ds=None):
    """eg3d_torgb_small_bwd: the toRGB data gradient for small pixel counts (+ the producing layer's activation backward when act_bwd is an
    ActBwdSpec the kernel takes).  Returns None when nothing was launched (geometry not the kernel's), else True / False = the activation
    backward was / was not fused (as conv_igemm)."""
    assert is_cl(dy) and is_cl(x) and is_cl(dx)
    n, c, h, w = x.shape
    p = L.TorgbSmallBwdParams(dy=dy.data_ptr(), wa=wa.data_ptr(), s=styles.data_ptr(), xin=x.data_ptr(), addend=addend.data_ptr() if addend is not None else None,
                              dx=dx.data_ptr(), ds=ds.data_ptr() if ds is not None else None, out_amax=out_amax.data_ptr() if out_amax is not None else None,
                              N=n, H=h, W=w, C=c, Cp=dy.shape[1], ldg=dy.shape[1], ldx=c, wa_row=wa.stride(0), act_on=0,
                              no_mid=0 if TORGB_MID_BWD else 1)          # (the library picks the streaming form itself from 4096 pixels: the switch must reach it)
    if addend_scale is not None:          # the addend is an unfinished split-K data gradient: finish it in this launch (PendingDgrad)
        p.add_scale, p.add_ds = addend_scale.data_ptr(), addend_ds.data_ptr() if addend_ds is not None else None
    fused = False
    if act_bwd is not None:
        p.act_on = 1
        act_bwd.fill(p.act_bwd)
        fused = bool(L.lib().eg3d_torgb_small_bwd_supported(C.byref(p)))
        if not fused:
            p.act_on = 0
            p.act_bwd = L.ActBwd()
            p.out_amax = None
    if not fused and not L.lib().eg3d_torgb_small_bwd_supported(C.byref(p)):
        return None
    if n * h * w > TORGB_SMALL_BWD_MAX_PIX and not (TORGB_MID_BWD and L.lib().eg3d_torgb_mid_bwd_supported(C.byref(p))):
        return None
    L.check(L.lib().eg3d_torgb_small_bwd(C.byref(p), L.stream_ptr()), 'torgb_small_bwd')
    return fused


def conv_atomic(x, wp, Ck, Nc, out, classes, in_stride=1, out_stride=1, in_scale=None, ksplit=1, **igemm_kw):
    """A split (EPI_ATOMIC) launch of the implicit GEMM with `ksplit` slices into the pre-zeroed `out`."""
    conv_igemm(x, wp, Ck, Nc, out, classes, in_stride=in_stride, out_stride=out_stride, in_scale=in_scale, epi=L.EPI_ATOMIC, ksplit=ksplit, **igemm_kw)


class HipAdam:
    """torch.optim.Adam(params, lr, betas, eps) for fp32 leaves in one launch per 32 leaves (`eg3d_adam_step`), with two extras the latent
    projector's step wants folded in: a second gradient per leaf (the noise regulariser's, which does not go through autograd) and the
    renormalisation of the noise maps after the update (w_projector.py:264-270).  The learning rate and the step count are device scalars, so
    a captured step can be replayed for every step index.  Same surface as the torch optimiser where the projector touches it
    (`param_groups[0]['lr']` -- a float or a device scalar --, `zero_grad`, `step`, `state`)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = list(params)
        dev = self.params[0].device
        self._lr_t = lr if torch.is_tensor(lr) else torch.tensor(float(lr), device=dev)
        self._lr_host = None if torch.is_tensor(lr) else float(lr)
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps)]
        self.step_t = torch.zeros((), device=dev)
        self.state = {p: dict(exp_avg=torch.zeros_like(p, memory_format=torch.contiguous_format),
                              exp_avg_sq=torch.zeros_like(p, memory_format=torch.contiguous_format), step=self.step_t) for p in self.params}
        for p in self.params:
            assert p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def step(self, extra_grads=None, normalize=None, skip=None):
        """extra_grads: {param: tensor} added to the gradient; normalize: {param: number of equal slices renormalised separately} (a noise
        map [N,1,r,r] of N independent images: N); skip: device scalar, != 0 -> this step changes nothing (eg3d_adam_list::skip)."""
        g = self.param_groups[0]
        lr = g['lr']
        if torch.is_tensor(lr):
            self._lr_t = lr
        elif getattr(self, '_lr_host', None) != float(lr):          # (a fill launch per step only when the host value changed)
            self._lr_t.fill_(float(lr))
            self._lr_host = float(lr)
        items = []
        for p in self.params:
            e = extra_grads.get(p) if extra_grads else None
            if p.grad is None and e is None:
                continue
            gr = p.grad
            assert gr is None or (gr.dtype == torch.float32 and gr.is_contiguous()), 'HipAdam: fp32 contiguous gradients'
            assert e is None or (e.dtype == torch.float32 and e.is_contiguous() and e.shape == p.shape)
            st = self.state[p]
            k = int(normalize.get(p, 0)) if normalize else 0
            parts = max(k, 1)
            n = p.numel() // parts
            assert n * parts == p.numel()
            for i in range(parts):
                o = 4 * n * i
                items.append((p.data_ptr() + o, gr.data_ptr() + o if gr is not None else None, e.data_ptr() + o if e is not None else None,
                              st['exp_avg'].data_ptr() + o, st['exp_avg_sq'].data_ptr() + o, n, 1 if k else 0))
        dev = self.params[0].device
        # the kernel writes through raw pointers: tell autograd / memo() that the leaves changed (no launch involved)
        touched = [p for p in self.params if p.grad is not None or (extra_grads and extra_grads.get(p) is not None)]
        bump = getattr(torch._C._autograd, '_unsafe_set_version_counter', None)
        if bump is not None and touched:
            bump(touched, [p._version + 1 for p in touched])
        else:
            weights_changed()
        for lo in range(0, len(items), L.ADAM_ITEMS_MAX):
            bank = items[lo:lo + L.ADAM_ITEMS_MAX]
            a = L.AdamList(n=len(bank), bump_step=int(lo + L.ADAM_ITEMS_MAX >= len(items)), beta1=g['betas'][0], beta2=g['betas'][1], eps=g['eps'],
                           lr=self._lr_t.data_ptr(), step=self.step_t.data_ptr(), skip=skip.data_ptr() if skip is not None else None)
            for j, it in enumerate(bank):
                a.items[j] = L.AdamItem(*it)
            ws = zeros((2 * len(bank) + 1,), dev)
            L.check(L.lib().eg3d_adam_step(C.byref(a), L.ptr(ws), L.stream_ptr()), 'adam_step')


def early_stop_flag(value, threshold, done):
    """done <- 1 where value <= threshold (sticky device flag; eg3d_early_stop_flag)."""
    L.check(L.lib().eg3d_early_stop_flag(L.ptr(value), float(threshold), L.ptr(done), L.stream_ptr()), 'early_stop_flag')
    return done


def noise_normalize_(bufs):
    for lo in range(0, len(bufs), NOISE_BANK_MAX):
        n, xs, res = _buf_arrays(bufs[lo:lo + NOISE_BANK_MAX])
        ws = zeros((2 * n,), bufs[0].device)          # per-buffer (sum, sum of squares): multi-block path
        L.check(L.lib().eg3d_noise_normalize(xs, res, n, L.ptr(ws), L.stream_ptr()), 'noise_normalize')
