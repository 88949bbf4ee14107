"""conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter): 2-D convolution with optional FIR-filtered
up- / down-sampling -- the operator surface of the reference's torch_utils/ops/conv2d_resample.py:48-143, composed from the gfx950
implicit-GEMM conv and FIR ops.  The weight flip of the reference's `_conv2d_wrapper` (:38-39) is a tap re-indexing inside the conv
kernel's tap list here, never a tensor copy.

Routing (same decompositions as the reference, so results agree term by term):
    no resampling                 conv with symmetric padding; any other padding goes through a pad-only FIR pass first
    down only                     1x1: FIR+decimate, then conv;  kxk: FIR, then conv with stride `down`
    up only, 1x1                  conv, then zero-insert + FIR (gain up^2)
    up (kxk, any down)            transposed conv with stride `up` (never a zero-stuffed input), FIR (gain up^2) [, decimate]
"""
import torch

from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _filter_size, _padding


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """Correlation (flip_weight=True, torch's convention) or true convolution of x with w; the spatial flip is `_flip_taps`."""
    flip_taps = (not flip_weight) and w.shape[2] * w.shape[3] > 1
    op = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return op(x, w, stride=stride, padding=padding, groups=groups, _flip_taps=flip_taps)


def _fir_margin(taps: int, up: int, down: int):
    """(low, high) padding a `taps`-wide FIR needs so that the resampled grid stays centred on the input grid."""
    lo = hi = 0
    if up > 1:
        lo, hi = lo + (taps + up - 1) // 2, hi + (taps - up) // 2
    if down > 1:
        lo, hi = lo + (taps - down + 1) // 2, hi + (taps - down) // 2
    return lo, hi


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    if not (isinstance(x, torch.Tensor) and x.ndim == 4 and isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype):
        raise TypeError('conv2d_resample: x [N,C,H,W] and w [O,I,kh,kw] of the same dtype expected')
    if f is not None and not (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32):
        raise TypeError('conv2d_resample: f must be a float32 1-D / 2-D FIR filter (upfirdn2d.setup_filter)')
    if not (isinstance(up, int) and isinstance(down, int) and up >= 1 and down >= 1):
        raise ValueError('conv2d_resample: up / down must be positive integers')
    kh, kw = int(w.shape[2]), int(w.shape[3])
    fw, fh = _filter_size(f)
    base = _padding(padding)
    mx, my = _fir_margin(fw, up, down), _fir_margin(fh, up, down)
    px0, px1, py0, py1 = base[0] + mx[0], base[1] + mx[1], base[2] + my[0], base[3] + my[1]
    conv = lambda t, wt=w, **kw_: _conv(t, wt, groups=groups, **{'flip_weight': flip_weight, **kw_})     # noqa: E731
    fir = lambda t, **kw_: upfirdn2d.upfirdn2d(t, f, flip_filter=flip_filter, **kw_)                    # noqa: E731
    pointwise = kh == 1 and kw == 1

    if up == 1 and down == 1:
        if px0 == px1 >= 0 and py0 == py1 >= 0:
            return conv(x, padding=[py0, px0])
        return conv(upfirdn2d.upfirdn2d(x, None, padding=[px0, px1, py0, py1], flip_filter=flip_filter))      # crop / asymmetric pad only
    if up == 1:
        if pointwise:
            return conv(fir(x, down=down, padding=[px0, px1, py0, py1]))
        return conv(fir(x, padding=[px0, px1, py0, py1]), stride=down)
    if pointwise and down == 1:
        return fir(conv(x), up=up, padding=[px0, px1, py0, py1], gain=up * up)
    # k x k with up > 1: stride-`up` transposed conv produces the (up*H + k - up)-sized grid the FIR then trims to up*H
    if groups == 1:
        wt = w.transpose(0, 1)
    else:       # [g * O/g, I/g, kh, kw] -> the transposed layout [g * I/g, O/g, kh, kw], group by group (conv2d_resample.py:118-121)
        co, cig = w.shape[0], w.shape[1]
        wt = w.reshape(groups, co // groups, cig, kh, kw).transpose(1, 2).reshape(groups * cig, co // groups, kh, kw)
    qx0, qx1, qy0, qy1 = px0 - (kw - 1), px1 - (kw - up), py0 - (kh - 1), py1 - (kh - up)
    tx, ty = max(min(-qx0, -qx1), 0), max(min(-qy0, -qy1), 0)           # the part of a negative FIR padding the transposed conv can crop itself
    y = conv(x, wt, stride=up, padding=[ty, tx], transpose=True, flip_weight=not flip_weight)
    y = fir(y, padding=[qx0 + tx, qx1 + tx, qy0 + ty, qy1 + ty], gain=up * up)
    return fir(y, down=down) if down > 1 else y
