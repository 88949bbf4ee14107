"""bias_act -- same public surface as the reference's torch_utils/ops/bias_act.py:23-88 (activation_funcs table,
bias_act(x, b, dim, act, alpha, gain, clamp, impl)), executed by eg3d_bias_act on gfx950.

First- and second-order gradients are provided (the reference contract: bias_act.py:128-209); derivatives are keyed on
the output like the native kernel (bias_act.cu:76,145).  impl='cuda' (default) raises on CPU tensors -- no automatic fallback; an
explicit impl='ref' selects the plain-torch composite of _ref_impl.py (reference: bias_act.py:84-88)."""
import math
from types import SimpleNamespace

import torch

from ... import _lib as L
from ... import hipops as H

activation_funcs = {
    'linear':   SimpleNamespace(def_alpha=0.0, def_gain=1.0,          cuda_idx=1, ref='',  has_2nd_grad=False),
    'relu':     SimpleNamespace(def_alpha=0.0, def_gain=math.sqrt(2), cuda_idx=2, ref='y', has_2nd_grad=False),
    'lrelu':    SimpleNamespace(def_alpha=0.2, def_gain=math.sqrt(2), cuda_idx=3, ref='y', has_2nd_grad=False),
    'tanh':     SimpleNamespace(def_alpha=0.0, def_gain=1.0,          cuda_idx=4, ref='y', has_2nd_grad=True),
    'sigmoid':  SimpleNamespace(def_alpha=0.0, def_gain=1.0,          cuda_idx=5, ref='y', has_2nd_grad=True),
    'elu':      SimpleNamespace(def_alpha=0.0, def_gain=1.0,          cuda_idx=6, ref='y', has_2nd_grad=True),
    'selu':     SimpleNamespace(def_alpha=0.0, def_gain=1.0,          cuda_idx=7, ref='y', has_2nd_grad=True),
    'softplus': SimpleNamespace(def_alpha=0.0, def_gain=1.0,          cuda_idx=8, ref='y', has_2nd_grad=True),
    'swish':    SimpleNamespace(def_alpha=0.0, def_gain=math.sqrt(2), cuda_idx=9, ref='x', has_2nd_grad=True),
}


def _dense(t):
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        return t
    return t.contiguous()


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, spec, alpha, gain, clamp):
        x = _dense(x)
        bb = b.contiguous() if b is not None else None
        trivial = spec.cuda_idx == 1 and gain == 1 and clamp < 0 and bb is None
        y = x if trivial else H.bias_act_raw(x, bb, None, None, None, 0, dim, spec.cuda_idx, alpha, gain, clamp)
        keep_x = 'x' in spec.ref or spec.has_2nd_grad
        # y is also kept whenever a clamp is active: the reference's native path drops it for 'linear'/'swish' and thereby ignores
        # the clamp in the gradient (bias_act.py:153-156 + bias_act.cu:145); the `_ref` path -- the parity anchor -- honours it.
        ctx.save_for_backward(x if keep_x else None, bb if keep_x else None, y if ('y' in spec.ref or clamp >= 0) else None)
        ctx.cfg = (dim, spec, alpha, gain, clamp, bb is not None, x.dim())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        dim, spec, alpha, gain, clamp, has_b, nd = ctx.cfg
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if spec.cuda_idx != 1 or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, dim, spec, alpha, gain, clamp)
        if has_b and ctx.needs_input_grad[1]:
            db = dx.sum([i for i in range(nd) if i != dim])
        return dx, db, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, dim, spec, alpha, gain, clamp):
        ref = y if y is not None else (x if x is not None else dy)
        dy = dy.contiguous(memory_format=torch.channels_last) if (ref.dim() == 4 and ref.stride(1) == 1 and ref.shape[1] > 1) else dy.contiguous()
        dx = H.bias_act_raw(dy, b, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp)
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        ctx.cfg = (dim, spec, alpha, gain, clamp)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        dim, spec, alpha, gain, clamp = ctx.cfg
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, dim, spec, alpha, gain, clamp)
        if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = H.bias_act_raw(d_dx.contiguous() if d_dx.stride() != dy.stride() else d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
            if b is not None and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.dim()) if i != dim])
        return d_dy, d_x, d_b, None, None, None, None, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b) * gain); shape/dtype/layout of x are preserved."""
    assert isinstance(x, torch.Tensor)
    assert impl in ('ref', 'cuda')
    spec = activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if b is not None:
        assert b.dim() == 1 and 0 <= dim < x.dim() and b.shape[0] == x.shape[dim]
        b = b.to(x.dtype)
    if impl == 'ref':       # explicit request only (reference: bias_act.py:84-88); there is no automatic fallback to it
        from ._ref_impl import bias_act_ref
        return bias_act_ref(x, b, dim, act, alpha, gain, clamp)
    L.require_cuda(x, b)
    return _BiasAct.apply(x, b, dim, spec, alpha, gain, clamp)
