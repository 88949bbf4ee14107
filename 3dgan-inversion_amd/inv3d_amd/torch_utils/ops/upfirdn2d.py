"""upfirdn2d -- public surface of the reference's torch_utils/ops/upfirdn2d.py (setup_filter :72, upfirdn2d :120,
filter2d :279, upsample2d :315, downsample2d :354), executed by eg3d_upfirdn2d on gfx950 (NCHW or channels_last,
fp16/fp32/fp64).  Gradients of arbitrary order: the backward is the same op with up<->down (upfirdn2d.py:253-271)."""
import numpy as np
import torch

from ... import _lib as L
from ... import hipops as H


def _pair(v, what):
    x, y = (v, v) if isinstance(v, int) else v
    if int(x) < 1 or int(y) < 1:
        raise ValueError(f'{what} factors must be >= 1')
    return int(x), int(y)


def _scaling(s):
    return _pair(s, 'up / down')


def _padding(p):
    """int | [px, py] | [px0, px1, py0, py1]  ->  (px0, px1, py0, py1)."""
    vals = [int(p)] * 4 if isinstance(p, int) else [int(v) for v in p]
    if len(vals) == 2:
        vals = [vals[0], vals[0], vals[1], vals[1]]
    if len(vals) != 4:
        raise ValueError('padding must have 1, 2 or 4 entries')
    return tuple(vals)


def _filter_size(f):
    """(width, height) of a FIR filter; None is the identity (1 tap)."""
    if f is None:
        return 1, 1
    if not (isinstance(f, torch.Tensor) and f.ndim in (1, 2)):
        raise TypeError('filter must be a 1-D or 2-D tensor')
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Filter taps -> the float32 tensor the FIR ops take (reference contract: upfirdn2d.py:72-116).  A 1-D filter with fewer than 8
    taps is expanded to its outer product unless `separable` says otherwise (longer ones stay 1-D and run as two passes); `normalize`
    scales to unit DC gain; `gain` is split evenly over the dimensions of a separable filter (gain^(ndim/2))."""
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if taps.ndim > 2 or taps.numel() == 0:
        raise ValueError('filter must be a scalar, 1-D or 2-D and non-empty')
    taps = taps.reshape(1) if taps.ndim == 0 else taps
    keep_1d = (taps.ndim == 1 and taps.numel() >= 8) if separable is None else bool(separable)
    if taps.ndim == 1 and not keep_1d:
        taps = taps[:, None] * taps[None, :]
    if taps.ndim != (1 if keep_1d else 2):
        raise ValueError('separable=True needs a 1-D filter')
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = taps.flip(tuple(range(taps.ndim)))
    return (taps * (gain ** (taps.ndim / 2))).to(device=device)


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, pad, flip, gain):
        assert x.ndim == 4
        upx, upy = up
        dnx, dny = down
        px0, px1, py0, py1 = pad
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        if f.ndim == 1 and f.shape[0] == 1:
            f = f.square().unsqueeze(0)
        if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
            x = x.contiguous()
        if f.ndim == 2:
            y = H.upfirdn2d_raw(x, f, upx, upy, dnx, dny, px0, px1, py0, py1, flip, gain)
        else:   # separable: horizontal then vertical pass, gain on the second (upfirdn2d.py:245-247)
            y = H.upfirdn2d_raw(x, f.unsqueeze(0), upx, 1, dnx, 1, px0, px1, 0, 0, flip, 1.0)
            y = H.upfirdn2d_raw(y, f.unsqueeze(1), 1, upy, 1, dny, 0, 0, py0, py1, flip, gain)
        ctx.save_for_backward(f)
        ctx.cfg = (x.shape, up, down, pad, flip, gain)
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        (_, _, ih, iw), (upx, upy), (dnx, dny), (px0, px1, py0, py1), flip, gain = ctx.cfg
        _, _, oh, ow = dy.shape
        fw, fh = _filter_size(f)
        p = (fw - px0 - 1, iw * upx - ow * dnx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * dny + py0 - upy + 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _Upfirdn2d.apply(dy, f, (dnx, dny), (upx, upy), p, not flip, gain)
        return dx, None, None, None, None, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Pad, upsample, FIR-filter and downsample a batch of 2-D images."""
    assert isinstance(x, torch.Tensor) and impl in ('ref', 'cuda')
    if impl == 'ref':       # explicit request only (reference: upfirdn2d.py:160-164); there is no automatic fallback to it
        from ._ref_impl import upfirdn2d_ref
        return upfirdn2d_ref(x, f, _scaling(up), _scaling(down), _padding(padding), bool(flip_filter), float(gain))
    L.require_cuda(x, f)
    return _Upfirdn2d.apply(x, f, _scaling(up), _scaling(down), _padding(padding), bool(flip_filter), float(gain))


def _centred(f, padding, lo_fn, hi_fn):
    """User padding + the margins that keep the output grid centred: lo_fn / hi_fn map (taps, axis) to the low / high margin."""
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    return [px0 + lo_fn(fw, 0), px1 + hi_fn(fw, 0), py0 + lo_fn(fh, 1), py1 + hi_fn(fh, 1)]


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """FIR-filter at the same resolution ('same' output size for padding=0; reference: upfirdn2d.py:279-311)."""
    p = _centred(f, padding, lambda t, a: t // 2, lambda t, a: (t - 1) // 2)
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Zero-insert by `up` and FIR-filter; the gain is multiplied by up_x * up_y so that magnitudes are preserved (:315-350)."""
    u = _scaling(up)
    p = _centred(f, padding, lambda t, a: (t + u[a] - 1) // 2, lambda t, a: (t - u[a]) // 2)
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * u[0] * u[1], impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """FIR-filter and keep every `down`-th sample (:354-389)."""
    d = _scaling(down)
    p = _centred(f, padding, lambda t, a: (t - d[a] + 1) // 2, lambda t, a: (t - d[a]) // 2)
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
