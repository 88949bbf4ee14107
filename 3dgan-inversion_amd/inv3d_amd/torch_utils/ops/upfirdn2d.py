"""upfirdn2d -- public surface of the reference's torch_utils/ops/upfirdn2d.py (setup_filter :72, upfirdn2d :120,
filter2d :279, upsample2d :315, downsample2d :354), executed by eg3d_upfirdn2d on gfx950 (NCHW or channels_last,
fp16/fp32/fp64).  Gradients of arbitrary order: the backward is the same op with up<->down (upfirdn2d.py:253-271)."""
import numpy as np
import torch

from ... import _lib as L
from ... import hipops as H


def _scaling(s):
    if isinstance(s, int):
        s = [s, s]
    sx, sy = s
    assert sx >= 1 and sy >= 1
    return int(sx), int(sy)


def _padding(p):
    if isinstance(p, int):
        p = [p, p]
    p = [int(v) for v in p]
    if len(p) == 2:
        p = [p[0], p[0], p[1], p[1]]
    return tuple(p)


def _filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2)
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """FIR setup: 1-D filters with < 8 taps become their outer product; unit DC gain; optional flip / gain."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert f.ndim in (0, 1, 2) and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


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
    if impl == 'ref':
        raise NotImplementedError("impl='ref' is not part of the MI355X product path; see oracle/ (tests only)")
    L.require_cuda(x, f)
    return _Upfirdn2d.apply(x, f, _scaling(up), _scaling(down), _padding(padding), bool(flip_filter), float(gain))


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    upx, upy = _scaling(up)
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    dx, dy = _scaling(down)
    px0, px1, py0, py1 = _padding(padding)
    fw, fh = _filter_size(f)
    p = [px0 + (fw - dx + 1) // 2, px1 + (fw - dx) // 2, py0 + (fh - dy + 1) // 2, py1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
