"""conv2d / conv_transpose2d with the reference's conv2d_gradfix surface (torch_utils/ops/conv2d_gradfix.py:23-45,
flags `enabled`, `weight_gradients_disabled`, context manager `no_weight_gradients`), executed by the fp32-MFMA implicit
GEMM of libeg3d_hip.so.  Forward, data gradient (the opposite-transpose op, :139-143) and weight gradient (:166-173) are
three launches of the same family; gradients of gradients re-enter these Functions.

Supported: fp32, kernels <= 3x3, conv2d with any isotropic stride and symmetric padding (stride > 1 = the data gradient of the
transposed conv, as the reference states the duality), conv_transpose2d with isotropic stride and symmetric padding (cropping),
output_padding < stride, groups (one launch per group), dilated stride-1 k x k kernels (other tap offsets for the tap-list kernel), any channel
counts (padded to multiples of 4 on the contraction side).  Dilated AND strided kernels, dilated transposed convs and anisotropic strides raise
NotImplementedError -- there is no silent fallback."""
import contextlib

import torch

from ... import _lib as L
from ... import hipops as H

enabled = False                      # kept for API parity; this implementation is always the one used on GPU tensors
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _pad_c4(t):
    """Pad the channel dim of an NCHW-shaped tensor with zeros to a multiple of 4 (16-byte pixels)."""
    c = t.shape[1]
    if c % 4 == 0:
        return t
    return torch.cat([t, t.new_zeros(t.shape[0], 4 - c % 4, *t.shape[2:])], 1)


class _ConvFn(torch.autograd.Function):
    """mode 'corr': y = correlate(x, w) stride 1, padding p.   mode 'convT': y = conv_transpose(x, w^T-layout [I,O,kh,kw]) stride up."""

    @staticmethod
    def forward(ctx, x, w, mode, up, pad, flip_taps):
        L.require_cuda(x, w)
        if x.dtype != torch.float32:
            raise NotImplementedError('eg3d conv kernels are fp32')
        xin = H.to_cl(_pad_c4(x))
        N, Cip, Hi, Wi = xin.shape
        if mode == 'corr':
            Co, Ci, kh, kw = w.shape
            wfull = w
        else:
            Ci, Co, kh, kw = w.shape
            wfull = w.transpose(0, 1)                  # [O,I,kh,kw] view: out[o, up*a+ky] += x[i,a] * w[i,o,ky]
        def _pack(wfull=wfull):
            wp = torch.cat([wfull, wfull.new_zeros(Co, Cip - Ci, kh, kw)], 1) if Cip != Ci else wfull
            return H.pack_weight_fwd(wp.detach())
        wf = H.memo(('gradfix_fwd', mode, Cip), [w], _pack)
        Cop = (Co + 3) // 4 * 4
        if mode == 'corr':
            Ho, Wo = Hi + 2 * pad[0] - up * (kh - 1), Wi + 2 * pad[1] - up * (kw - 1)         # mode 'corr': `up` carries the dilation
            if pad[0] != pad[1]:
                raise NotImplementedError('asymmetric conv padding')
            if Ho < 1 or Wo < 1:
                raise ValueError('conv2d: kernel larger than the padded input')
            cls = H.classes_corr(Ho, Wo, kh, kw, pad[0], flip_taps, dil=up)
            out = (H.zeros_cl if Cop != Co else H.empty_cl)(N, Cop, Ho, Wo, x.device)
            H.conv_igemm(xin, wf, Cip, Co, out, cls, epi=L.EPI_STORE)
        else:
            cls, Ho, Wo = H.classes_convT(Hi, Wi, kh, kw, up, flip_taps)
            out = H.zeros_cl(N, Cop, Ho, Wo, x.device)       # phases without taps (k < up) stay zero
            H.conv_igemm(xin, wf, Cip, Co, out, cls, out_stride=up, epi=L.EPI_STORE)
        ctx.save_for_backward(x, w)
        ctx.cfg = (mode, up, pad, flip_taps, Co, Ci)
        y = out[:, :Co] if Cop != Co else out
        if mode == 'convT' and (pad[0] or pad[1]):
            y = y[:, :, pad[0]: y.shape[2] - pad[0], pad[1]: y.shape[3] - pad[1]]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        mode, up, pad, flip_taps, Co, Ci = ctx.cfg
        dx = dw = None
        if mode == 'convT' and (pad[0] or pad[1]):
            dy = torch.nn.functional.pad(dy, [pad[1], pad[1], pad[0], pad[0]])
        if ctx.needs_input_grad[0]:
            dx = _ConvDataGradFn.apply(dy, w, mode, up, pad, flip_taps, x.shape)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _ConvWeightGradFn.apply(dy, x, mode, up, pad, flip_taps, w.shape)
        return dx, dw, None, None, None, None


class _ConvDataGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, w, mode, up, pad, flip_taps, x_shape):
        g = H.to_cl(_pad_c4(dy.float()))
        N, Cgp, Hg, Wg = g.shape
        _, Ci, Hi, Wi = x_shape
        if mode == 'corr':
            Co, _, kh, kw = w.shape
            wfull = w
        else:
            _, Co, kh, kw = w.shape
            wfull = w.transpose(0, 1)
        def _pack(wfull=wfull):
            wp = torch.cat([wfull, wfull.new_zeros(Cgp - Co, Ci, kh, kw)], 0) if Cgp != Co else wfull
            return H.pack_weight_adj(wp.detach())
        wa = H.memo(('gradfix_adj', mode, Cgp), [w], _pack)                   # [Ci, taps*Cgp]
        Cip = (Ci + 3) // 4 * 4
        dx = (H.zeros_cl if Cip != Ci else H.empty_cl)(N, Cip, Hi, Wi, dy.device)
        if mode == 'corr':
            cls = H.classes_corr_adjoint(Hi, Wi, kh, kw, pad[0], flip_taps, dil=up)
            H.conv_igemm(g, wa, Cgp, Ci, dx, cls, epi=L.EPI_STORE)
        else:
            cls = H.classes_convT_adjoint(Hi, Wi, kh, kw, up, flip_taps)
            H.conv_igemm(g, wa, Cgp, Ci, dx, cls, in_stride=up, epi=L.EPI_STORE)
        ctx.save_for_backward(dy, w)
        ctx.cfg = (mode, up, pad, flip_taps, x_shape)
        return dx[:, :Ci] if Cip != Ci else dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, w = ctx.saved_tensors
        mode, up, pad, flip_taps, x_shape = ctx.cfg
        d_dy = d_w = None
        if ctx.needs_input_grad[0]:
            y = _ConvFn.apply(d_dx, w, mode, up, (0, 0) if mode == 'convT' else pad, flip_taps)
            d_dy = y
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            d_w = _ConvWeightGradFn.apply(dy, d_dx, mode, up, pad, flip_taps, w.shape)
        return d_dy, d_w, None, None, None, None, None


class _ConvWeightGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, mode, up, pad, flip_taps, w_shape):
        g = H.to_cl(_pad_c4(dy.float()))
        xin = H.to_cl(_pad_c4(x.float()))
        N, Cip, Hi, Wi = xin.shape
        if mode == 'corr':
            Co, Ci, kh, kw = w_shape
            cls = H.classes_corr(g.shape[2], g.shape[3], kh, kw, pad[0], flip_taps, dil=up)
            out_stride = 1
        else:
            Ci, Co, kh, kw = w_shape
            cls = H.classes_convT(Hi, Wi, kh, kw, up, flip_taps)[0]
            out_stride = up
        dwp = torch.zeros((Co, kh * kw * Cip), device=dy.device)
        H.conv_wgrad(xin, g, Cip, Co, dwp, cls, in_stride=1, out_stride=out_stride)
        dw = dwp.view(Co, kh, kw, Cip)[..., :Ci].permute(0, 3, 1, 2)          # [O,I,kh,kw]
        if mode == 'convT':
            dw = dw.transpose(0, 1)
        ctx.save_for_backward(dy, x)
        ctx.cfg = (mode, up, pad, flip_taps, w_shape)
        return dw.contiguous()

    @staticmethod
    def backward(ctx, d_dw):
        dy, x = ctx.saved_tensors
        mode, up, pad, flip_taps, w_shape = ctx.cfg
        d_dy = d_x = None
        if ctx.needs_input_grad[0]:
            y = _ConvFn.apply(x, d_dw, mode, up, (0, 0) if mode == 'convT' else pad, flip_taps)
            d_dy = y
        if ctx.needs_input_grad[1]:
            d_x = _ConvDataGradFn.apply(dy, d_dw, mode, up, pad, flip_taps, x.shape)
        return d_dy, d_x, None, None, None, None, None


def _check(weight, dilation):
    d = _pair(dilation)
    if d[0] != d[1]:
        raise NotImplementedError('anisotropic dilation')
    if weight.shape[2] * weight.shape[3] > 9:
        raise NotImplementedError('kernels larger than 9 taps')
    return d[0]


def _grouped(fn, input, weight, groups, w_out_dim):
    """groups > 1 (torch_utils/ops/conv2d_gradfix.py:37-45 hands every combination to ATen): one launch per group on channel slices --
    the generator never takes this path (its per-sample weights are expressed as activation scaling, fused.ModConvLayerFn)."""
    cin = input.shape[1] // groups
    wpg = weight.shape[0] // groups
    outs = [fn(input[:, g * cin:(g + 1) * cin], weight[g * wpg:(g + 1) * wpg]) for g in range(groups)]
    return torch.cat(outs, 1)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, _flip_taps=False):
    """F.conv2d's contract (torch_utils/ops/conv2d_gradfix.py:37-40): any stride, symmetric padding, dilation, groups; <= 9 taps."""
    dil = _check(weight, dilation)
    s, p = _pair(stride), _pair(padding)
    if s[0] != s[1]:
        raise NotImplementedError('anisotropic stride')
    if groups != 1:
        if input.shape[1] % groups or weight.shape[0] % groups:
            raise ValueError('channels must be divisible by groups')
        y = _grouped(lambda x_, w_: conv2d(x_, w_, None, stride, padding, dilation, 1, _flip_taps), input, weight, groups, 0)
    elif dil != 1:
        # a dilated k x k kernel is a (d (k - 1) + 1)^2 kernel with zeros between the taps: for the tap-list kernel that is only a change
        # of the tap offsets (loader-split kernel: any offset, reads outside the image are zero)
        kh, kw = weight.shape[2:]
        if kh * kw == 1:
            y = conv2d(input, weight, None, stride, padding, 1, 1, _flip_taps)
        elif s == (1, 1):        # the tap-list kernel only sees other tap offsets (the Function's `up` slot carries the dilation in 'corr' mode)
            y = _ConvFn.apply(input, weight, 'corr', dil, p, _flip_taps)
        else:
            raise NotImplementedError('dilated AND strided k x k convolution (no caller on or near the inversion path)')
    elif s == (1, 1):
        y = _ConvFn.apply(input, weight, 'corr', 1, p, _flip_taps)
    else:
        # stride-s correlation = the data gradient of the stride-s transposed convolution with the same weight tensor read as [in_t, out_t]
        # (conv2d_gradfix.py:139-143 states the same duality): reuse that Function, whose own backward is the transposed conv + weight gradient
        if p != (0, 0):
            input = torch.nn.functional.pad(input, [p[1], p[1], p[0], p[0]])
        kh, kw = weight.shape[2:]
        ho, wo = (input.shape[2] - kh) // s[0] + 1, (input.shape[3] - kw) // s[0] + 1
        if ho < 1 or wo < 1:
            raise ValueError('conv2d: kernel larger than the padded input')
        need = ((ho - 1) * s[0] + kh, (wo - 1) * s[0] + kw)             # the rows / columns the strided taps actually reach
        input = input[:, :, :need[0], :need[1]]
        y = _ConvDataGradFn.apply(input, weight, 'convT', s[0], (0, 0), _flip_taps, (input.shape[0], weight.shape[0], ho, wo))
    return y if bias is None else y + bias.reshape(1, -1, 1, 1)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1, _flip_taps=False):
    """F.conv_transpose2d's contract (conv2d_gradfix.py:42-45) for isotropic stride, symmetric padding, dilation 1; output_padding < stride."""
    if _check(weight, dilation) != 1:
        raise NotImplementedError('dilated transposed convolution')
    s, op, p = _pair(stride), _pair(output_padding), _pair(padding)
    if s[0] != s[1]:
        raise NotImplementedError('anisotropic stride')
    if op[0] >= max(s[0], 1) or op[1] >= max(s[1], 1) or min(op) < 0:
        raise ValueError('output_padding must be smaller than the stride')
    if groups != 1:
        if input.shape[1] % groups or weight.shape[0] % groups:
            raise ValueError('channels must be divisible by groups')
        y = _grouped(lambda x_, w_: conv_transpose2d(x_, w_, None, stride, padding, output_padding, 1, 1, _flip_taps), input, weight, groups, 1)
    elif op == (0, 0):
        y = _ConvFn.apply(input, weight, 'convT', s[0], p, _flip_taps)
    else:
        # output_padding only changes WHICH rows / columns of the un-cropped result are kept: `padding` fewer at the top / left,
        # `padding - output_padding` fewer at the bottom / right (rows no tap reaches are zero)
        full = _ConvFn.apply(input, weight, 'convT', s[0], (0, 0), _flip_taps)
        grow = [max(op[1] - p[1], 0), max(op[0] - p[0], 0)]
        if grow[0] or grow[1]:
            full = torch.nn.functional.pad(full, [0, grow[0], 0, grow[1]])
        Hf, Wf = full.shape[2], full.shape[3]
        y = full[:, :, p[0]: Hf - max(p[0] - op[0], 0), p[1]: Wf - max(p[1] - op[1], 0)]
    return y if bias is None else y + bias.reshape(1, -1, 1, 1)
