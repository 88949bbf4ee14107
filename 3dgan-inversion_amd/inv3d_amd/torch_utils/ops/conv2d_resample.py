"""conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter) -- reference surface
torch_utils/ops/conv2d_resample.py:48-143, composed from the gfx950 conv and FIR ops.  The weight flip of
`_conv2d_wrapper` (:38-39) is a tap re-indexing inside the kernel's tap list, never a tensor copy."""
import torch

from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _filter_size, _padding


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    kh, kw = w.shape[2], w.shape[3]
    flip_taps = (not flip_weight) and (kw > 1 or kh > 1)
    if transpose:
        return conv2d_gradfix.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups, _flip_taps=flip_taps)
    return conv2d_gradfix.conv2d(x, w, stride=stride, padding=padding, groups=groups, _flip_taps=flip_taps)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4 and isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    cout, cin_g, kh, kw = w.shape
    fw, fh = _filter_size(f)
    px0, px1, py0, py1 = _padding(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2; py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2; py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2

    if kw == 1 and kh == 1 and down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)
    if kw == 1 and kh == 1 and up > 1 and down == 1:
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:
        if groups != 1:
            raise NotImplementedError('grouped up-sampling convolution')
        wt = w.transpose(0, 1)
        px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    x = upfirdn2d.upfirdn2d(x, (f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x
