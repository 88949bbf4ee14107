"""filtered_lrelu on gfx950: bias -> up-sampling FIR -> leaky ReLU * gain, clamp -> down-sampling FIR as one HIP launch
(`eg3d_filtered_lrelu`), with gradients of any order.

Same call signature and semantics as the reference operator (`torch_utils/ops/filtered_lrelu.py:41-120`, semantics defined by its
`_filtered_lrelu_ref`, :123-155).  StyleGAN3's alias-free layers are its only user upstream; EG3D's StyleGAN2 backbone never reaches
it, so it is exported for API completeness of the operator library rather than for the hot path.

Structure (differs from the reference's sign-bit plugin chain): the forward writes an fp32 *derivative mask* d act/d t at the
up-sampled resolution; every gradient is then the linear operator  v -> FIR2(mask * FIR1(v))  -- the same kernel in "mask" mode --
whose transpose is again such an operator with the two FIR stages swapped and flipped, so one small autograd.Function closes the
whole tower of higher-order gradients.
"""
import math

import numpy as np
import torch

from ... import _lib as L


def _filter_hw(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    return (f.shape[0], f.shape[0]) if f.ndim == 1 else (f.shape[0], f.shape[1])


def _dense(f, device):
    """None | 1-D separable | 2-D filter -> dense 2-D fp32 tensor on `device` (or None)."""
    if f is None:
        return None
    f = f.to(device=device, dtype=torch.float32)
    if f.ndim == 1:
        f = torch.outer(f, f)
    return f.contiguous()


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = [int(v) for v in padding]
    assert len(padding) in (2, 4)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    return padding


class _Cfg:
    """Static description of one launch: y = FIR_down(f2, q) o middle o FIR_up(f1, p)."""
    __slots__ = ('up', 'down', 'p', 'q', 'gain1', 'gain2', 'flip1', 'flip2', 'in_hw', 'mid_hw', 'out_hw')

    def __init__(self, up, down, p, q, gain1, gain2, flip1, flip2, in_hw, f1_hw, f2_hw):
        self.up, self.down, self.p, self.q, self.gain1, self.gain2, self.flip1, self.flip2 = up, down, list(p), list(q), gain1, gain2, flip1, flip2
        h, w = in_hw
        self.in_hw = (h, w)
        self.mid_hw = (h * up + p[2] + p[3] - (f1_hw[0] - 1), w * up + p[0] + p[1] - (f1_hw[1] - 1))
        mh, mw = self.mid_hw
        self.out_hw = ((mh + q[2] + q[3] - f2_hw[0] + down) // down, (mw + q[0] + q[1] - f2_hw[1] + down) // down)

    def transposed(self, f1_hw, f2_hw):
        """Configuration of the adjoint operator (input = this one's output grid).  Padding rules of the up/down-sampling FIR adjoint:
        torch_utils/ops/upfirdn2d.py:262-269."""
        (h, w), (mh, mw), (oh, ow) = self.in_hw, self.mid_hw, self.out_hw
        # adjoint of the second stage (filter f2, down) becomes the new first stage (up = down)
        p_new = [f2_hw[1] - self.q[0] - 1, mw - ow * self.down + self.q[0], f2_hw[0] - self.q[2] - 1, mh - oh * self.down + self.q[2]]
        # adjoint of the first stage (filter f1, up) becomes the new second stage (down = up)
        q_new = [f1_hw[1] - self.p[0] - 1, w * self.up - mw + self.p[0] - self.up + 1, f1_hw[0] - self.p[2] - 1, h * self.up - mh + self.p[2] - self.up + 1]
        t = _Cfg(self.down, self.up, p_new, q_new, self.gain2, self.gain1, not self.flip2, not self.flip1, (oh, ow), f2_hw, f1_hw)
        assert t.mid_hw == self.mid_hw and t.out_hw == self.in_hw, (t.mid_hw, self.mid_hw, t.out_hw, self.in_hw)
        return t


def _launch(x, b, f1, f2, mask, cfg, mode, act=(1.0, 0.0, -1.0)):
    L.require_cuda(x)
    if x.dtype not in (torch.float32, torch.float16):
        raise L.Eg3dHipError(f'filtered_lrelu: unsupported dtype {x.dtype}')
    x = x.contiguous()
    n, c, h, w = x.shape
    assert (h, w) == cfg.in_hw
    y = torch.empty((n, c, *cfg.out_hw), dtype=x.dtype, device=x.device)
    p = L.FlreluParams()
    p.x, p.y = x.data_ptr(), y.data_ptr()
    p.b = b.contiguous().data_ptr() if b is not None else None
    p.fu = f1.data_ptr() if f1 is not None else None
    p.fd = f2.data_ptr() if f2 is not None else None
    p.mask = mask.data_ptr() if mask is not None else None
    p.dtype = L.F32 if x.dtype == torch.float32 else L.F16
    p.N, p.C, p.H, p.W = n, c, h, w
    p.fuh, p.fuw = (f1.shape if f1 is not None else (1, 1))
    p.fdh, p.fdw = (f2.shape if f2 is not None else (1, 1))
    p.up, p.down = cfg.up, cfg.down
    p.px0, p.px1, p.py0, p.py1 = cfg.p
    p.qx0, p.qx1, p.qy0, p.qy1 = cfg.q
    p.Ho, p.Wo = cfg.out_hw
    p.flip_fu, p.flip_fd, p.mode = int(cfg.flip1), int(cfg.flip2), mode
    p.gain1, p.gain2 = float(cfg.gain1), float(cfg.gain2)
    p.gain, p.slope, p.clamp = (float(v) for v in act)
    import ctypes as C
    L.check(L.lib().eg3d_filtered_lrelu(C.byref(p), L.stream_ptr()), 'filtered_lrelu')
    return y


class _MaskedFirFn(torch.autograd.Function):
    """v -> FIR2(mask * FIR1(v)); linear in v.  Its gradient is the same Function on the transposed configuration."""

    @staticmethod
    def forward(ctx, v, mask, f1, f2, cfg):
        ctx.cfg, ctx.f = cfg, (f1, f2)
        ctx.save_for_backward(mask)
        return _launch(v, None, f1, f2, mask, cfg, mode=1)

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        f1, f2 = ctx.f
        dv = None
        if ctx.needs_input_grad[0]:
            dv = _MaskedFirFn.apply(dy, mask, f2, f1, ctx.cfg.transposed(_filter_hw(f1), _filter_hw(f2)))
        return dv, None, None, None, None


class _FilteredLReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, f1, f2, cfg, act):
        n, c = x.shape[:2]
        mask = torch.empty((n, c, *cfg.mid_hw), dtype=torch.float32, device=x.device)
        y = _launch(x, b, f1, f2, mask, cfg, mode=0, act=act)
        ctx.cfg, ctx.f, ctx.has_b = cfg, (f1, f2), b is not None
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        f1, f2 = ctx.f
        dx = db = None
        if ctx.needs_input_grad[0] or (ctx.has_b and ctx.needs_input_grad[1]):
            dx = _MaskedFirFn.apply(dy, mask, f2, f1, ctx.cfg.transposed(_filter_hw(f1), _filter_hw(f2)))
            if ctx.has_b and ctx.needs_input_grad[1]:
                db = dx.sum([0, 2, 3])
        return (dx if ctx.needs_input_grad[0] else None), db, None, None, None, None


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False,
                   impl='cuda'):
    """See the module docstring; arguments as the reference's `filtered_lrelu()` (filtered_lrelu.py:41-120):
    x [N,C,H,W] fp32/fp16, fu/fd 1-D separable or 2-D FIR filters (None = identity), b [C] bias, integer up/down factors,
    padding = int | [x, y] | [x_before, x_after, y_before, y_after] in the up-sampled grid, gain/slope/clamp of the leaky ReLU."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert impl in ('ref', 'cuda')
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    assert gain == float(gain) and gain > 0 and slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype and tuple(b.shape) == (x.shape[1],)
    if impl == 'ref':       # explicit request only (reference: filtered_lrelu.py:113-120); there is no automatic fallback to it
        from ._ref_impl import filtered_lrelu_ref
        return filtered_lrelu_ref(x, fu, fd, b, up, down, _pad4(padding), float(gain), float(slope), clamp, bool(flip_filter))
    f1, f2 = _dense(fu, x.device), _dense(fd, x.device)
    cfg = _Cfg(up, down, _pad4(padding), [0, 0, 0, 0], float(up) ** 2, 1.0, bool(flip_filter), bool(flip_filter), x.shape[2:],
               _filter_hw(f1), _filter_hw(f2))
    if min(cfg.mid_hw) < 1 or min(cfg.out_hw) < 1:
        raise L.Eg3dHipError(f'filtered_lrelu: empty output {cfg.mid_hw} -> {cfg.out_hw}')
    act = (float(gain), float(slope), -1.0 if clamp is None else float(clamp))
    return _FilteredLReluFn.apply(x, b, f1, f2, cfg, act)


def filtered_lrelu_act_(x, si=None, sx=0, sy=0, gain=math.sqrt(2.0), slope=0.2, clamp=None, write_signs=False):
    """The plugin's stand-alone in-place activation (filtered_lrelu.cpp:217-272; what the reference's generic fallback calls between its two
    upfirdn2d passes, filtered_lrelu.py:225-231).  x: contiguous [N,C,H,W] fp32 / fp16, modified in place.  write_signs=True: returns the
    packed 2-bit sign image [N,C,H,ceil(W/4)] uint8; si given: applies a recorded sign image at offset (sx, sy) (the backward); neither:
    the plain forward.  Returns the sign image (new, given, or None)."""
    L.require_cuda(x, si)
    if not (x.dim() == 4 and x.is_contiguous() and x.dtype in (torch.float32, torch.float16)):
        raise L.Eg3dHipError('filtered_lrelu_act_: contiguous [N,C,H,W] fp32 / fp16 tensor expected')
    n, c, h, w = x.shape
    mode = 0
    if write_signs:
        si = torch.empty((n, c, h, (w + 3) // 4), dtype=torch.uint8, device=x.device)
        mode = 1
    elif si is not None:
        if not (si.dtype == torch.uint8 and si.dim() == 4 and si.is_contiguous() and si.shape[:2] == x.shape[:2]):
            raise L.Eg3dHipError('filtered_lrelu_act_: signs must be a contiguous uint8 [N,C,sH,sW/4] tensor')
        mode = 2
    sh, sw = (si.shape[2], si.shape[3] * 4) if si is not None else (0, 0)
    dt = L.F32 if x.dtype == torch.float32 else L.F16
    L.check(L.lib().eg3d_filtered_lrelu_act(x.data_ptr(), L.ptr(si), dt, n * c, h, w, sh, sw, int(sx), int(sy), float(gain), float(slope),
                                            float(-1.0 if clamp is None else clamp), mode, L.stream_ptr()), 'filtered_lrelu_act')
    return si
