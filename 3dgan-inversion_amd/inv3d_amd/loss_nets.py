"""Perceptual-loss networks of the inversion loops on the gfx950 kernels.

The reference scores every step with third-party networks whose code and weights live outside its tree:

  VGG16LPIPS      <- the torchscript `vgg16.pt` of stylegan2-ada-pytorch, called as vgg16(img_0_255, resize_images=False,
                     return_lpips=True) (training/projectors/w_projector.py:50-52,112,215-219): LPIPS-VGG features as one flat vector
                     per image whose squared distance is the LPIPS distance.
  VGG16Features   <- torchvision.models.vgg16().features, children 0..14 = conv3_3 without its ReLU
                     (training/projectors/w_projector.py:55-58, training/warping_loss.py:74-105 with layers='14').
  LPIPSAlex       <- lpips.LPIPS(net='alex') (training/coaches/base_coach.py:48,111-112).

None of the three packages / checkpoints exists offline, so the networks are restated here from their published definitions
(Simonyan & Zisserman 2015 configuration D; Krizhevsky 2014 as shipped by torchvision; Zhang et al. 2018, lpips v0.1:
ScalingLayer, normalize_tensor with eps 1e-10, non-negative 1x1 `lin` layers, spatial mean, sum over layers) with the state-dict
keys of the original modules, so real checkpoints load with `load_state_dict`; without them the weights are seeded random
(constructor argument `seed`).  The CPU restatement used by the tests is oracle/loss_nets_oracle.py; parity against the third-party packages
themselves is unpinned.

All convolutions run on the implicit-GEMM kernel with bias + ReLU fused in its epilogue (frozen weights: forward and data gradient
only), pooling and the LPIPS head on csrc/loss_ops.hip.  Activations are fp32 channels-last; the 3-channel image is carried padded
to 4 channels.  There is no fallback: tensors must live on the GPU."""
import math
from ctypes import byref as C_byref
from typing import List, Optional, Sequence, Tuple

import os
import torch

from . import _lib as L
from . import hipops as H
from .fused import _auto_ksplit
from .torch_utils.ops import bias_act

WGRAD_PRECISION = os.environ.get('EG3D_POSE_WGRAD', 'f16x3')        # weight gradients of the in-loop pose estimator: 'f16x3' | 'f32' (v_mfma_f32_32x32x2_f32)
# Frozen loss networks (VGG16-LPIPS, VGG16 features, AlexNet-LPIPS): when the operand range is known -- the producing layer's epilogue reports
# max|out| (a device scalar; one reduction pass for the network input and after split-K layers), the activation-backward pass reports max|dz| --
# the convs run in the generator's arithmetic (two-piece fp16 operands range-normalised by that maximum, three MFMA products per fp32
# product) instead of bf16x6 (six): same fp32-equivalent result, half the matrix work.  EG3D_LOSS_NET_F16X3=0: bf16x6 everywhere.
LOSS_NET_F16X3 = os.environ.get('EG3D_LOSS_NET_F16X3', '1') != '0'
LOSS_NET_PRECISION = 'bf16x6'        # fp32-equivalent for any operand range (pixel values up to 255 enter these networks)


# ----------------------------------------------------------------------------------------------------------- tap lists
def _chunks(taps, n=9):
    return [taps[i:i + n] for i in range(0, len(taps), n)]


def _classes_strided(Ho, Wo, kh, kw, pad):
    """out[y,x] = sum_k w[ky,kx] * in[y*s + ky - pad, x*s + kx - pad]; kernels with more than 9 taps become several classes that
    accumulate into the same output grid."""
    taps = [(ky - pad, kx - pad, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
    return [H._mk_class(Ho, Wo, 0, 0, c) for c in _chunks(taps)]


def _classes_strided_adjoint(Hi, Wi, kh, kw, s, pad):
    """Data gradient of _classes_strided: dx[a*s+py, b*s+px] = sum over taps with ky = py+pad (mod s) of g[a + (py+pad-ky)/s, ...] *
    w[ky,kx]: one class per output phase (py,px), chunked to 9 taps.  Returns (classes, overlapping)."""
    cls, overlapping = [], False
    for py in range(s):
        for px in range(s):
            taps = []
            for ky in range(kh):
                if (py + pad - ky) % s:
                    continue
                for kx in range(kw):
                    if (px + pad - kx) % s:
                        continue
                    taps.append(((py + pad - ky) // s, (px + pad - kx) // s, ky * kw + kx))
            Ha, Wa = (Hi - py + s - 1) // s, (Wi - px + s - 1) // s
            if not taps or Ha <= 0 or Wa <= 0:
                overlapping = True          # a phase nobody writes: the caller must start from zeros
                continue
            ch = _chunks(taps)
            overlapping |= len(ch) > 1
            cls += [H._mk_class(Ha, Wa, py, px, c) for c in ch]
    return cls, overlapping


def _launch_groups(x, wp, Ck, Nc, out, classes, accumulate, **kw):
    """At most 4 classes per launch.  accumulate: several classes add into the same pixels (out pre-zeroed, atomics)."""
    for i in range(0, len(classes), 4):
        H.conv_igemm(x, wp, Ck, Nc, out, classes[i:i + 4], epi=L.EPI_ATOMIC if accumulate else L.EPI_STORE, **dict(dict(precision=LOSS_NET_PRECISION), **kw))


LOSS_NET_PRESPLIT = os.environ.get('EG3D_LOSS_NET_PRESPLIT', '1') != '0'          # 3x3 stride-1 layers of the frozen loss networks on the pre-split kernels (conv_v3 / conv_ws) instead of the loader-split one


def _presplit_plan(Ck, Nc, classes, N, H_, W_):
    """How a frozen 3x3 / stride-1 / pad-1 layer (Ck -> Nc channels on an H_ x W_ grid) runs on the pre-split kernels: ('v3', (rows, waves)) -- the wave-split
    kernel with its fused bias + activation epilogue (csrc/conv_v3.hip), ('ws', None) -- the weight-streaming kernel for <= 256 cells (csrc/conv_ws.hip,
    accumulates into a zeroed buffer), or None (loader-split implicit GEMM).  Round 6: VGG16 at 256^2 spends 1.0 of its 1.45 ms in thirteen 30 - 50 us launches
    of the loader-split kernel at 50 - 130 TFLOP/s plus a bias pass and an absmax pass per layer."""
    if not LOSS_NET_PRESPLIT or len(classes) != 1 or classes[0].ntaps != 9 or Ck % 16 or H.CONV_MODE != 'auto' or not H.USE_V2:
        return None
    if Nc % 32 == 0 and N * H_ * W_ >= 256 and H.conv_ws_ok(Ck, Nc, classes, N, H_, W_):          # (16^2: VGG16 conv5_x; smaller grids -- LPIPS-Alex at 128^2 -- stay where they were)
        return 'ws', None
    if Nc % 64 or not H.USE_V3 or N * H_ * W_ < 1024:
        return None
    wg4 = N * -(-H_ // 4) * -(-W_ // 32) * (Nc // 64)
    return 'v3', ((4, 4) if wg4 >= 192 else (2, 8))


class _ConvActFn(torch.autograd.Function):
    """y = act(conv2d(x, w, stride, pad) + b).  x, y: fp32 channels-last, channel counts multiples of 4.  The loss networks call it
    with frozen w, b (forward + data gradient); with trainable w / b (the pose estimator, pose_net.py) the weight gradient comes from
    eg3d_conv2d_wgrad_f32 over the same tap classes and the bias gradient is the pixel sum of the pre-activation gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, alpha=0.0, gain=1.0):
        L.require_cuda(x, weight, bias)
        assert H.is_cl(x) and x.dtype == torch.float32
        N, Cip, Hi, Wi = x.shape
        Co, Ci, kh, kw = weight.shape
        assert Ci <= Cip and Cip % 4 == 0 and Co % 4 == 0, (weight.shape, x.shape)
        trainable = ctx.needs_input_grad[1]

        def _pack():
            wp = torch.cat([weight, weight.new_zeros(Co, Cip - Ci, kh, kw)], 1) if Cip != Ci else weight
            return H.pack_weight_fwd(wp.detach().float())
        wf = _pack() if trainable else H.memo(('lossnet_fwd', Cip), [weight], _pack)      # trainable weights are new tensors every step
        Ho, Wo = (Hi + 2 * pad - kh) // stride + 1, (Wi + 2 * pad - kw) // stride + 1
        cls = _classes_strided(Ho, Wo, kh, kw, pad)
        ks = _auto_ksplit(cls, N, Co, Cip) if len(cls) == 1 else 1
        f16 = LOSS_NET_F16X3 and not trainable and act in ('linear', 'relu', 'lrelu')
        pk = {}
        if f16:             # operand range: the producer's report if the tensor carries one, else one reduction pass (hipops.amax_of)
            pk = dict(precision='f16x3', a_amax=H.amax_of(x), w_pieces=H.memo(('lossnet_fwd_pieces', Cip), [weight], lambda: H.split_weight_pieces(wf)))
        y_amax = None
        route = _presplit_plan(Cip, Co, cls, N, Ho, Wo) if (f16 and kh == 3 and kw == 3 and stride == 1 and pad == 1) else None
        if route is not None and route[0] == 'v3':
            wimg = H.memo(('lossnet_fwd_split', Cip), [weight], lambda: H.split_weight(wf, Co, Cip, 9))
            y = H.empty_cl(N, Co, Ho, Wo, x.device)
            y_amax = H.zeros((1,), x.device)
            H.conv_v3(H.split_activation(x, H.amax_of(x)), wimg, y, cls, plan=route[1], epi=L.EPI_FWD, bias=bias, act=act, alpha=alpha, gain=gain, out_amax=y_amax)
        elif route is not None:
            wimg = H.memo(('lossnet_fwd_split', Cip), [weight], lambda: H.split_weight(wf, Co, Cip, 9))
            z = H.zeros_cl(N, Co, Ho, Wo, x.device)
            H.conv_ws(x, wimg, z, cls, x_amax=H.amax_of(x))
            y = H.bias_act_raw(z, bias, None, None, None, 0, 1, L.ACT_IDS[act], alpha, gain, -1.0)
        elif len(cls) == 1 and ks == 1:
            y = H.empty_cl(N, Co, Ho, Wo, x.device)
            y_amax = H.zeros((1,), x.device) if f16 else None
            H.conv_igemm(x, wf, Cip, Co, y, cls, in_stride=stride, epi=L.EPI_FWD, bias=bias, act=act, alpha=alpha, gain=gain, out_amax=y_amax,
                         **(pk or dict(precision=LOSS_NET_PRECISION)))
        else:           # several tap classes per pixel, or a grid too small to fill the chip (split-K): accumulate, then bias + act
            z = H.zeros_cl(N, Co, Ho, Wo, x.device)
            _launch_groups(x, wf, Cip, Co, z, cls, True, in_stride=stride, ksplit=ks, **pk)
            y = H.bias_act_raw(z, bias, None, None, None, 0, 1, L.ACT_IDS[act], alpha, gain, -1.0)
        if y_amax is not None:
            H.tag_amax(y, y_amax)          # (read by the next layer's amax_of; max pooling passes it on)
        ctx.save_for_backward(y, weight, x if trainable else None)
        ctx.cfg = (stride, pad, act, x.shape, (Ho, Wo), float(alpha), float(gain))
        ctx.f16 = f16
        return y

    @staticmethod
    def backward(ctx, dy):
        y, weight, x = ctx.saved_tensors
        stride, pad, act, (N, Cip, Hi, Wi), (Ho, Wo), alpha, gain = ctx.cfg
        Co, Ci, kh, kw = weight.shape
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        if not (need_x or need_w or need_b):
            return (None,) * 8
        dy = H.to_cl(dy.float())
        dx = dw = db = None
        amax = None
        if need_w and WGRAD_PRECISION != 'f32' and act in ('linear', 'relu', 'lrelu') and Co % 4 == 0 and Co <= 1024:
            # trainable weights (the pose estimator's 7x7 stem): activation backward, bias gradient and max|dz| (the operand range of the
            # f16x3 weight gradient) from one pass
            db = H.zeros((Co,), dy.device) if need_b else None
            amax = H.zeros((1,), dy.device)
            dz = H.epilogue_bwd(dy, y, H.empty_cl(N, Co, Ho, Wo, dy.device), act=act, alpha=alpha, gain=gain, dbias=db, dz_amax=amax)
        elif ctx.f16 and need_x and not (act == 'linear' and gain == 1.0) and Co % 4 == 0 and Co <= 1024:
            amax = H.zeros((1,), dy.device)       # frozen network: the activation backward also reports max|dz|, the data gradient's operand range
            dz = H.epilogue_bwd(dy, y, H.empty_cl(N, Co, Ho, Wo, dy.device), act=act, alpha=alpha, gain=gain, dz_amax=amax)
        else:
            dz = dy if (act == 'linear' and gain == 1.0) else H.bias_act_raw(dy, None, None, y, None, 1, 1, L.ACT_IDS[act], alpha, gain, -1.0)
        if need_x:
            def _pack():
                wp = torch.cat([weight, weight.new_zeros(Co, Cip - Ci, kh, kw)], 1) if Cip != Ci else weight
                return H.pack_weight_adj(wp.detach().float())                 # [Cip, taps*Co]
            wa = _pack() if need_w else H.memo(('lossnet_adj', Cip), [weight], _pack)
            cls, overlapping = _classes_strided_adjoint(Hi, Wi, kh, kw, stride, pad)
            ks = _auto_ksplit(cls, N, Cip, Co) if len(cls) == 1 else 1
            overlapping |= ks > 1
            dx = (H.zeros_cl if overlapping else H.empty_cl)(N, Cip, Hi, Wi, dy.device)
            pk = {}
            if ctx.f16 and amax is not None and not need_w:
                pk = dict(precision='f16x3', a_amax=amax, w_pieces=H.memo(('lossnet_adj_pieces', Cip), [weight], lambda: H.split_weight_pieces(wa)))
            route = _presplit_plan(Co, Cip, cls, N, Hi, Wi) if (pk and kh == 3 and kw == 3 and stride == 1 and pad == 1 and len(cls) == 1) else None
            if route is not None:
                wimg = H.memo(('lossnet_adj_split', Cip), [weight], lambda: H.split_weight(wa, Cip, Co, 9))
                if route[0] == 'v3':
                    if overlapping:
                        dx = H.empty_cl(N, Cip, Hi, Wi, dy.device)        # (split-K was planned for the loader-split kernel: the wave-split one writes every pixel once)
                    H.conv_v3(H.split_activation(dz, amax), wimg, dx, cls, plan=route[1], epi=L.EPI_STORE)
                else:
                    if not overlapping:
                        dx = H.zeros_cl(N, Cip, Hi, Wi, dy.device)
                    H.conv_ws(dz, wimg, dx, cls, x_amax=amax)
            else:
                _launch_groups(dz, wa, Co, Cip, dx, cls, overlapping, out_stride=stride, ksplit=ks, **pk)
        if need_w:
            dwp = H.zeros((Co, kh * kw * Cip), dy.device)
            cls = _classes_strided(Ho, Wo, kh, kw, pad)
            for i in range(0, len(cls), 4):
                if amax is not None:
                    H.conv_wgrad(x, dz, Cip, Co, dwp, cls[i:i + 4], in_stride=stride, out_stride=1, precision=WGRAD_PRECISION, g_amax=amax)
                else:
                    H.conv_wgrad(x, dz, Cip, Co, dwp, cls[i:i + 4], in_stride=stride, out_stride=1)
            dw = dwp.view(Co, kh, kw, Cip)[..., :Ci].permute(0, 3, 1, 2).contiguous()
        if need_b and db is None:
            db = dz.sum((0, 2, 3))
        return dx, dw, db, None, None, None, None, None


class _ConvScaleActFn(torch.autograd.Function):
    """y = act(conv2d(x, w * a[o], stride, pad) + b) with trainable w, a, b -- a conv followed by an eval-mode BatchNorm (a = gamma /
    sqrt(var + eps) folded into the weights, b the shifted bias) and ReLU, as the in-loop pose estimator evaluates it every step.  Per
    layer and step: ONE kernel builds both packed images of the folded weight (eg3d_pack_conv_weight_scaled), ONE kernel turns the packed
    weight gradient into dw (parameter layout) and da (eg3d_unpack_weight_grad), and the activation backward also sums the bias gradient
    -- instead of ~12 weight-sized ATen passes (fold, two permute-copies, un-permute, two products, two reductions, fills)."""

    @staticmethod
    def forward(ctx, x, weight, a, bias, stride, pad, act, packed=None):
        L.require_cuda(x, weight, a, bias)
        assert H.is_cl(x) and x.dtype == torch.float32
        N, Ci, Hi, Wi = x.shape
        Co, Ciw, kh, kw = weight.shape
        assert Ciw == Ci and Ci % 4 == 0 and Co % 4 == 0 and kh * kw <= 9, (weight.shape, x.shape)
        w = weight.detach().contiguous().float()
        av = a.detach().contiguous().float()
        T = kh * kw
        if packed is not None:          # both images of w * a[o] were built with every other layer's in one launch (pose_net.ResNetPose.forward)
            wf, wa = packed
        else:
            wf = torch.empty((Co, T * Ci), device=x.device)
            wa = torch.empty((Ci, T * Co), device=x.device)
            L.check(L.lib().eg3d_pack_conv_weight_scaled(w.data_ptr(), av.data_ptr(), wf.data_ptr(), wa.data_ptr(), Co, Ci, T, L.stream_ptr()), 'pack_conv_weight_scaled')
        Ho, Wo = (Hi + 2 * pad - kh) // stride + 1, (Wi + 2 * pad - kw) // stride + 1
        cls = _classes_strided(Ho, Wo, kh, kw, pad)
        ks = _auto_ksplit(cls, N, Co, Ci)
        b = bias.detach().contiguous().float()
        if ks == 1:
            y = H.empty_cl(N, Co, Ho, Wo, x.device)
            H.conv_igemm(x, wf, Ci, Co, y, cls, in_stride=stride, epi=L.EPI_FWD, bias=b, act=act, gain=1.0, precision=LOSS_NET_PRECISION)
        else:
            z = H.zeros_cl(N, Co, Ho, Wo, x.device)
            _launch_groups(x, wf, Ci, Co, z, cls, True, in_stride=stride, ksplit=ks)
            y = H.bias_act_raw(z, b, None, None, None, 0, 1, L.ACT_IDS[act], 0.0, 1.0, -1.0)
        ctx.save_for_backward(y, w, av, x, wa)
        ctx.cfg = (stride, pad, act, (Ho, Wo), kh, kw)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, w, av, x, wa = ctx.saved_tensors
        stride, pad, act, (Ho, Wo), kh, kw = ctx.cfg
        N, Ci, Hi, Wi = x.shape
        Co = w.shape[0]
        need_x, need_w, need_a, need_b = ctx.needs_input_grad[:4]
        dy = H.to_cl(dy.float())
        db = None
        if act == 'linear' and not ((need_w or need_a) and WGRAD_PRECISION != 'f32'):
            dz = dy
            if need_b:
                db = dy.sum((0, 2, 3))
            amax = None
        else:                                  # activation backward, the bias gradient and max|dz| (the weight gradient's operand range) in one pass
            db = H.zeros((Co,), dy.device) if need_b else None
            amax = H.zeros((1,), dy.device) if ((need_w or need_a) and WGRAD_PRECISION != 'f32') else None      # max|dz|: range of the weight gradient's fp16 operand
            dz = H.epilogue_bwd(dy, y, H.empty_cl(N, Co, Ho, Wo, dy.device), act=act, gain=1.0, dbias=db, dz_amax=amax)
        dx = dw = da = None
        if need_x:
            cls, overlapping = _classes_strided_adjoint(Hi, Wi, kh, kw, stride, pad)
            ks = _auto_ksplit(cls, N, Ci, Co) if len(cls) == 1 else 1
            overlapping |= ks > 1
            dx = (H.zeros_cl if overlapping else H.empty_cl)(N, Ci, Hi, Wi, dy.device)
            _launch_groups(dz, wa, Co, Ci, dx, cls, overlapping, out_stride=stride, ksplit=ks)
        if need_w or need_a:
            T = kh * kw
            dwp = H.zeros((Co, T * Ci), dy.device)
            if amax is not None:        # two-piece fp16 operands, three products per fp32 product (the generator's arithmetic); dz range-normalised by max|dz|
                H.conv_wgrad(x, dz, Ci, Co, dwp, _classes_strided(Ho, Wo, kh, kw, pad), in_stride=stride, out_stride=1, precision=WGRAD_PRECISION, g_amax=amax)
            else:
                H.conv_wgrad(x, dz, Ci, Co, dwp, _classes_strided(Ho, Wo, kh, kw, pad), in_stride=stride, out_stride=1)
            dw = torch.empty_like(w)
            da = torch.empty_like(av) if need_a else None
            L.check(L.lib().eg3d_unpack_weight_grad(dwp.data_ptr(), w.data_ptr(), av.data_ptr(), dw.data_ptr(), da.data_ptr() if da is not None else None,
                                                    Co, Ci, Ci, T, L.stream_ptr()), 'unpack_weight_grad')
        return dx, (dw if need_w else None), da, db, None, None, None, None


def conv_scale_act_ok(x_channels, weight, act) -> bool:
    Co, Ci, kh, kw = weight.shape
    return x_channels == Ci and Ci % 4 == 0 and Co % 4 == 0 and kh * kw <= 9 and act in ('linear', 'relu')


def conv_scale_act(x, weight, a, bias, stride=1, pad=0, act='relu', packed=None):
    """act(conv2d(x, weight * a[:, None, None, None]) + bias): conv + folded eval-mode BatchNorm + activation, trainable (see _ConvScaleActFn);
    falls back to conv_act on a separately folded weight for shapes the fused form does not take (padded input channels, > 9 taps)."""
    Co, Ci, kh, kw = weight.shape
    if conv_scale_act_ok(x.shape[1], weight, act):
        return _ConvScaleActFn.apply(x, weight, a, bias, stride, pad, act, packed)
    return conv_act(x, weight * a.view(-1, 1, 1, 1), bias, stride, pad, act)


def conv_act(x, weight, bias, stride=1, pad=0, act='relu', alpha=0.0, gain=1.0):
    """act(conv2d(x, weight) + bias) * gain in one launch (act: 'linear' | 'relu' | 'lrelu' with slope alpha)."""
    return _ConvActFn.apply(x, weight, bias, stride, pad, act, alpha, gain)


# ---- the stand-in feature pyramid as direct convolutions (csrc/loss_ops.hip: eg3d_conv3x3_direct, eg3d_pool2_act_bwd) ----------------------------
DIRECT_PYRAMID = os.environ.get('EG3D_DIRECT_PYRAMID', '1') != '0'


def _direct_group(co: int, quads: int, ci: int) -> int:
    """Output channels per thread: the widest group that still leaves ~32 k threads (512 waves) for the chip (the kernel deals the
    input-channel quads to four waves of a block from 16 channels on)."""
    ks = 4 if ci >= 16 else 1
    for g in (4, 2, 1):
        if co % g == 0 and quads * (co // g) * ks >= 32768:
            return g
    return 1


def _pack_direct(w: torch.Tensor, g: int) -> torch.Tensor:
    """[Co,Ci,3,3] -> [Co/G][Ci/4][9][4][G] (Ci zero-padded to a multiple of 4)."""
    co, ci = w.shape[:2]
    cip = (ci + 3) // 4 * 4
    if cip != ci:
        w = torch.cat([w, w.new_zeros(co, cip - ci, 3, 3)], 1)
    return w.float().reshape(co // g, g, cip // 4, 4, 9).permute(0, 2, 4, 3, 1).contiguous()


def _conv3x3_direct(x, wp, co, g, *, y=None, pooled=None, act=False, alpha=0.2, gain=1.0, ga=None, gb=None):
    n, ci, h, w = x.shape
    p = L.Conv3x3DirectParams(x=x.data_ptr(), w=wp.data_ptr(), y=y.data_ptr() if y is not None else None,
                              pooled=pooled.data_ptr() if pooled is not None else None, N=n, H=h, W=w, Ci=ci, Co=co, G=g, act=1 if act else 0,
                              alpha=float(alpha), gain=float(gain), ga=ga.data_ptr() if ga is not None else None, gb=gb.data_ptr() if gb is not None else None)
    L.check(L.lib().eg3d_conv3x3_direct(C_byref(p), L.stream_ptr()), 'conv3x3_direct')


class _StubPyramidFn(torch.autograd.Function):
    """p_l = avg_pool2(lrelu(conv3x3(p_{l-1}, W_l)) * gain), l = 1..L, with frozen weights: one launch per level forward, two per level
    backward (pooling backward + sum of the level's two consumers' gradients + activation backward; then the data gradient), where the
    generic path took conv + pool forward and add + pool backward + activation backward + conv backward.  Returns every level's output."""

    @staticmethod
    def forward(ctx, x, alpha, gain, *weights):
        L.require_cuda(x, *weights)
        assert H.is_cl(x) and x.dtype == torch.float32
        ys, outs, cur = [], [], x
        for wt in weights:
            n, ci, h, w = cur.shape
            co = wt.shape[0]
            g = _direct_group(co, (h // 2) * (w // 2), ci)
            wp = H.memo(('direct_fwd', ci, g), [wt], lambda wt=wt, g=g: _pack_direct(wt.detach(), g))
            y = H.empty_cl(n, co, h, w, x.device)
            pl = H.empty_cl(n, co, h // 2, w // 2, x.device)
            _conv3x3_direct(cur, wp, co, g, y=y, pooled=pl, act=True, alpha=alpha, gain=gain)
            ys.append(y)
            outs.append(pl)
            cur = pl
        ctx.save_for_backward(*ys, *weights)
        ctx.cfg = (float(alpha), float(gain), len(weights), tuple(x.shape))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        alpha, gain, nl, xshape = ctx.cfg
        ys, weights = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        dx = None                                    # gradient arriving from the level above (w.r.t. this level's pooled output)
        for l in range(nl - 1, -1, -1):
            y, wt = ys[l], weights[l]
            n, co, h, w = y.shape
            ga = grads[l]
            if ga is not None:
                ga = H.to_cl(ga.float())
            if ga is None and dx is None:
                dx = None
                continue
            ci = xshape[1] if l == 0 else ys[l - 1].shape[1]
            g = _direct_group(ci, (h // 2) * (w // 2), co)
            wa = H.memo(('direct_adj', co, g), [wt], lambda wt=wt, g=g: _pack_direct(wt.detach().flip(2, 3).permute(1, 0, 2, 3), g))
            nxt = H.empty_cl(n, ci, h, w, y.device)
            # (dz formed inside the data-gradient launch -- eg3d_conv3x3_direct_params::ga -- measured 219.7 vs 220.1 steps/s twice: every
            #  channel-group block re-forms it; the separate pass stays)
            dz = H.empty_cl(n, co, h, w, y.device)
            L.check(L.lib().eg3d_pool2_act_bwd(ga.data_ptr() if ga is not None else None, dx.data_ptr() if dx is not None else None, y.data_ptr(), dz.data_ptr(),
                                               n, h, w, co, alpha, gain, L.stream_ptr()), 'pool2_act_bwd')
            _conv3x3_direct(dz, wa, ci, g, y=nxt)
            dx = nxt
        return (dx, None, None) + (None,) * nl


def stub_pyramid_ok(x: torch.Tensor, weights) -> bool:
    if not (DIRECT_PYRAMID and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0 and H.is_cl(x)):
        return False
    h, w, c = x.shape[2], x.shape[3], x.shape[1]
    for wt in weights:
        if wt.requires_grad or tuple(wt.shape[2:]) != (3, 3) or wt.shape[1] != c or wt.shape[0] % 4 or h % 2 or w % 2 or h < 2 or w < 2:
            return False
        h, w, c = h // 2, w // 2, wt.shape[0]
    return True


def stub_pyramid(x: torch.Tensor, weights, alpha: float = 0.2, gain: float = math.sqrt(2.0)):
    """The pooled outputs of every level of the stand-in pyramid (inversion.StubFeatureNet.stages) from the direct kernels."""
    return list(_StubPyramidFn.apply(x, float(alpha), float(gain), *weights))


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s):
        L.require_cuda(x)
        assert H.is_cl(x) and x.dtype == torch.float32
        N, C, Hi, Wi = x.shape
        Ho, Wo = (Hi - k) // s + 1, (Wi - k) // s + 1
        y = H.empty_cl(N, C, Ho, Wo, x.device)
        idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
        L.check(L.lib().eg3d_maxpool2d_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr() if idx is not None else None, N, Hi, Wi, C, C, k, s,
                                           L.stream_ptr()), 'maxpool2d_fwd')
        ctx.idx, ctx.cfg = idx, (k, s, x.shape)
        if getattr(x, '_eg3d_amax', None) is not None:
            H.tag_amax(y, x._eg3d_amax)          # max|pool(x)| <= max|x|: an upper bound is all the range normalisation needs
        return y

    @staticmethod
    def backward(ctx, dy):
        k, s, (N, C, Hi, Wi) = ctx.cfg
        dy = H.to_cl(dy.float())
        dx = H.empty_cl(N, C, Hi, Wi, dy.device)
        L.check(L.lib().eg3d_maxpool2d_bwd(dy.data_ptr(), ctx.idx.data_ptr(), dx.data_ptr(), N, Hi, Wi, C, C, k, s, L.stream_ptr()), 'maxpool2d_bwd')
        return dx, None, None


def max_pool(x, k, s):
    return _MaxPoolFn.apply(x, k, s)


class _LpipsHeadFn(torch.autograd.Function):
    """feat[n, off : off + H*W*C] = sqrt_lin[c] * x / (||x|| + eps) / sqrt(H*W)  for every tap of the trunk, one flat vector per image
    (pixel-major, channel-minor inside a layer's slice)."""

    @staticmethod
    def forward(ctx, eps, nscales, *args):
        eps, eps_inside = (eps[0], int(eps[1])) if isinstance(eps, tuple) else (eps, 0)
        xs, scales = args[:nscales], args[nscales:]
        N = xs[0].shape[0]
        sizes = [x.shape[1] * x.shape[2] * x.shape[3] for x in xs]
        F = sum(sizes)
        feat = torch.empty((N, F), device=xs[0].device, dtype=torch.float32)
        batch = L.UnitLevels(n=len(xs), N=N, eps_inside=eps_inside, eps=eps, feat_nstride=F)
        assert len(xs) <= L.UNIT_LEVELS_MAX
        off = 0
        for i, (x, sc, n) in enumerate(zip(xs, scales, sizes)):
            assert H.is_cl(x) and x.dtype == torch.float32 and x.shape[0] == N
            _, C, Hh, Ww = x.shape
            batch.levels[i] = L.UnitLevel(x=x.data_ptr(), scale=sc.data_ptr() if sc is not None else None, feat=feat.data_ptr() + 4 * off, dx=None,
                                          HW=Hh * Ww, C=C, ldx=C, mul=1.0 / math.sqrt(Hh * Ww))
            off += n
        L.check(L.lib().eg3d_unit_normalize_levels(C_byref(batch), 0, L.stream_ptr()), 'unit_normalize_levels')
        ctx.save_for_backward(*xs, *[s for s in scales if s is not None])
        ctx.cfg = (eps, nscales, sizes, F, eps_inside, [s is not None for s in scales])
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        eps, nscales, sizes, F, eps_inside, has_scale = ctx.cfg
        xs, rest = ctx.saved_tensors[:nscales], list(ctx.saved_tensors[nscales:])
        scales = [rest.pop(0) if h else None for h in has_scale]
        dfeat = dfeat.contiguous().float()
        grads, off, k = [], 0, 0
        batch = L.UnitLevels(N=xs[0].shape[0], eps_inside=eps_inside, eps=eps, feat_nstride=F)
        for i, (x, sc, n) in enumerate(zip(xs, scales, sizes)):
            if ctx.needs_input_grad[2 + i]:
                N, C, Hh, Ww = x.shape
                dx = H.empty_cl(N, C, Hh, Ww, x.device)
                batch.levels[k] = L.UnitLevel(x=x.data_ptr(), scale=sc.data_ptr() if sc is not None else None, feat=dfeat.data_ptr() + 4 * off,
                                              dx=dx.data_ptr(), HW=Hh * Ww, C=C, ldx=C, mul=1.0 / math.sqrt(Hh * Ww))
                k += 1
                grads.append(dx)
            else:
                grads.append(None)
            off += n
        if k:
            batch.n = k
            L.check(L.lib().eg3d_unit_normalize_levels(C_byref(batch), 1, L.stream_ptr()), 'unit_normalize_levels')
        return (None, None, *grads, *([None] * nscales))


def lpips_features(xs: Sequence[torch.Tensor], sqrt_lins: Sequence[torch.Tensor], eps: float = 1e-10) -> torch.Tensor:
    return _LpipsHeadFn.apply(eps, len(xs), *xs, *sqrt_lins)


class _ImagePrepFn(torch.autograd.Function):
    """(img4 [N,4,H,W] channels_last, 3 used channels) -> [N,4,H/f,W/f] channels_last:  mul * area-mean + add, channel 3 = 0."""

    @staticmethod
    def forward(ctx, img4, factor, mul, add):
        L.require_cuda(img4)
        assert H.is_cl(img4) and img4.shape[1] == 4 and img4.dtype == torch.float32
        N, _, Hh, Ww = img4.shape
        out = H.empty_cl(N, 4, Hh // factor, Ww // factor, img4.device)
        L.check(L.lib().eg3d_image_prepare_fwd(img4.data_ptr(), out.data_ptr(), N, Hh, Ww, factor, float(mul), float(add), L.stream_ptr()), 'image_prepare_fwd')
        ctx.cfg = (N, Hh, Ww, factor, float(mul))
        return out

    @staticmethod
    def backward(ctx, dout):
        N, Hh, Ww, factor, mul = ctx.cfg
        dout = H.to_cl(dout.float())
        dimg = H.empty_cl(N, 4, Hh, Ww, dout.device)
        L.check(L.lib().eg3d_image_prepare_bwd(dout.data_ptr(), dimg.data_ptr(), N, Hh, Ww, factor, mul, L.stream_ptr()), 'image_prepare_bwd')
        return dimg, None, None, None


def image_prepare(img4: torch.Tensor, factor: int, mul: float, add: float) -> torch.Tensor:
    """The feature networks' input from the generator's 4-float-pixel image in one pass: (img + 1) * 255/2 and the area resize of
    w_projector.py:198-200 (integer factor), channels kept padded to 4."""
    return _ImagePrepFn.apply(img4, int(factor), mul, add)


class _SqDistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        L.require_cuda(a, b)
        a, b = a.contiguous().float(), b.contiguous().float()
        N, F = a.shape
        out = H.zeros((N,), a.device)
        L.check(L.lib().eg3d_sqdist_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), N, F, L.stream_ptr()), 'sqdist_fwd')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        N, F = a.shape
        g = g.contiguous().float()
        da = torch.empty_like(a)
        L.check(L.lib().eg3d_sqdist_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), da.data_ptr(), N, F, L.stream_ptr()), 'sqdist_bwd')
        return (da if ctx.needs_input_grad[0] else None), (-da if ctx.needs_input_grad[1] else None)


def sqdist(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[N]: sum_i (a[n,i] - b[n,i])^2 of flat feature vectors [N,F] (F % 4 == 0), one launch per direction."""
    if a.shape[1] % 4 or a.data_ptr() % 16 or b.data_ptr() % 16:
        return (a - b).square().sum(1)
    return _SqDistFn.apply(a, b)


def _same_memory(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype == torch.float32


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class _ObjectiveFn(torch.autograd.Function):
    """total = sum_k group_weight[group_k] * scale_k * value_k and the per-group sums, one launch per term and direction."""

    @staticmethod
    def forward(ctx, spec, group_weights, *tensors):
        dev = tensors[0].device
        buf = torch.zeros((4 + len(group_weights),), device=dev)     # its own allocation (not the step's zero arena): callers keep the values
        total, parts = buf[:1], buf[4:]
        lib, st = L.lib(), L.stream_ptr()
        ti = 0
        for kind, group, scale in spec:
            part = parts.data_ptr() + 4 * group
            gw = float(group_weights[group]) * scale
            if kind == 'sq':
                a, b = tensors[ti], tensors[ti + 1]
                ti += 2
                L.check(lib.eg3d_sqdist_sum_fwd(a.data_ptr(), b.data_ptr(), a.numel(), part, scale, total.data_ptr(), gw, st), 'sqdist_sum_fwd')
            else:
                v = tensors[ti]
                ti += 1
                B, Hh, Ww = v.shape
                L.check(lib.eg3d_tv_norm_fwd(v.data_ptr(), B, Hh, Ww, part, scale, total.data_ptr(), gw, st), 'tv_norm_fwd')
        ctx.spec, ctx.gw = spec, group_weights
        ctx.save_for_backward(*tensors)
        ctx.mark_non_differentiable(parts)
        return total.view(()), parts

    @staticmethod
    def backward(ctx, g, _gparts):
        tensors = ctx.saved_tensors
        g = g.contiguous().float()
        lib, st = L.lib(), L.stream_ptr()
        grads, ti = [], 0
        for kind, group, scale in ctx.spec:
            gw = float(ctx.gw[group]) * scale
            if kind == 'sq':
                a, b = tensors[ti], tensors[ti + 1]
                da = None
                if ctx.needs_input_grad[2 + ti]:
                    da = torch.empty_like(a)               # same strides as a (dense): the kernels walk memory linearly
                    L.check(lib.eg3d_sqdist_sum_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), gw, da.data_ptr(), a.numel(), st), 'sqdist_sum_bwd')
                db = -da if (ctx.needs_input_grad[3 + ti] and da is not None) else None
                if ctx.needs_input_grad[3 + ti] and da is None:
                    db = torch.empty_like(b)
                    L.check(lib.eg3d_sqdist_sum_bwd(b.data_ptr(), a.data_ptr(), g.data_ptr(), gw, db.data_ptr(), a.numel(), st), 'sqdist_sum_bwd')
                grads += [da, db]
                ti += 2
            else:
                v = tensors[ti]
                dv = None
                if ctx.needs_input_grad[2 + ti]:
                    B, Hh, Ww = v.shape
                    dv = torch.empty_like(v)
                    L.check(lib.eg3d_tv_norm_bwd(v.data_ptr(), g.data_ptr(), gw, dv.data_ptr(), B, Hh, Ww, st), 'tv_norm_bwd')
                grads.append(dv)
                ti += 1
        return (None, None) + tuple(grads)


def weighted_objective(terms, group_weights):
    """(total, parts): total = sum_k group_weights[g_k] * scale_k * value_k, parts[g] = sum_{k in g} scale_k * value_k (not differentiable).
    terms: ('sq', group, scale, a, b) -- value = sum (a - b)^2 over two tensors of identical dense memory layout (numel % 4 == 0) -- or
    ('tv', group, scale, v) -- value = squared forward-difference total variation of a contiguous [B,H,W] map.  Returns None when an
    operand does not meet the kernels' layout requirements (the caller then composes the objective from ATen ops)."""
    spec, tensors = [], []
    for t in terms:
        if t[0] == 'sq':
            _, group, scale, a, b = t
            if not (a.is_cuda and _same_memory(a, b) and _dense(a) and a.numel() % 4 == 0 and a.numel() >= 4
                    and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
                return None
            tensors += [a, b]
        else:
            _, group, scale, v = t
            if not (v.is_cuda and v.dim() == 3 and v.is_contiguous() and v.dtype == torch.float32 and v.shape[1] >= 2 and v.shape[2] >= 2):
                return None
            tensors.append(v)
        spec.append((t[0], int(group), float(scale)))
    return _ObjectiveFn.apply(tuple(spec), tuple(float(w) for w in group_weights), *tensors)


def unit_features(xs: Sequence[torch.Tensor], eps: float = 1e-10) -> torch.Tensor:
    """[N, sum H*W*C]: per tap x * rsqrt(sum_c x^2 + eps) / sqrt(H*W), pixel-major inside a tap's slice (one launch per tap and direction)."""
    return _LpipsHeadFn.apply((eps, 1), len(xs), *xs, *([None] * len(xs)))


def _image_cl4(img: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, pre_mul: float, pre_add: float) -> torch.Tensor:
    """[N,3,H,W] -> channels-last [N,4,H,W]: ((img*pre_mul + pre_add) - shift) / scale in the first three channels, zero in the fourth."""
    n, c, h, w = img.shape
    assert c == 3, 'loss networks take RGB images'
    a = (pre_mul / scale).view(1, 3, 1, 1)
    b = ((pre_add - shift) / scale).view(1, 3, 1, 1)
    y = img.float() * a + b
    return torch.cat([y, y.new_zeros(n, 1, h, w)], 1).contiguous(memory_format=torch.channels_last)


# ----------------------------------------------------------------------------------------------------------- trunks
VGG16_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')      # torchvision cfg 'D'


class _VGG16Trunk(torch.nn.Module):
    """torchvision.models.vgg16().features with its module indices (conv k -> `<k>.weight`, `<k>.bias`); ReLU and pooling layers hold
    no parameters.  `run(x, upto, taps)` evaluates children 0..upto and returns the outputs of the children listed in `taps`."""

    def __init__(self):
        super().__init__()
        self.kinds = []
        cin = 3
        for v in VGG16_CFG:
            if v == 'M':
                self.kinds.append(('pool',))
            else:
                idx = len(self.kinds)
                conv = torch.nn.Conv2d(cin, v, 3, padding=1)
                conv.requires_grad_(False)
                self.add_module(str(idx), conv)
                self.kinds += [('conv', idx), ('relu',)]
                cin = v

    def run(self, x, upto, taps):
        outs = {}
        i = 0
        while i <= upto:
            kind = self.kinds[i]
            if kind[0] == 'conv':
                conv = getattr(self, str(i))
                fuse_relu = i + 1 <= upto and i not in taps
                x = conv_act(x, conv.weight, conv.bias, 1, 1, 'relu' if fuse_relu else 'linear')
                if i in taps:
                    outs[i] = x
                if fuse_relu:
                    i += 1                                   # the ReLU child has been applied in the conv epilogue
                    if i in taps:
                        outs[i] = x
            elif kind[0] == 'relu':                          # only reached when the conv output itself was tapped
                x = bias_act.bias_act(x, None, act='relu', gain=1)
                if i in taps:
                    outs[i] = x
            else:
                x = max_pool(x, 2, 2)
                if i in taps:
                    outs[i] = x
            i += 1
        return x, outs


def _he_init_(module: torch.nn.Module, seed: int):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() == 4 and 'lin' not in name:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(2.0 / fan_in))
            elif p.dim() == 4:                               # LPIPS lin layers: non-negative channel weights
                p.copy_(torch.rand(p.shape, generator=g) * (2.0 / p.shape[1]))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)


class _Lin(torch.nn.Module):
    """lpips.NetLinLayer: Dropout + Conv2d(C, 1, 1, bias=False); key `model.1.weight`."""

    def __init__(self, c):
        super().__init__()
        self.model = torch.nn.Sequential(torch.nn.Identity(), torch.nn.Conv2d(c, 1, 1, bias=False))
        self.model.requires_grad_(False)

    def sqrt_weight(self):
        w = self.model[1].weight
        return H.memo(('lpips_sqrt_lin',), [w], lambda: w.detach().float().clamp_min(0).sqrt().reshape(-1).contiguous())


LPIPS_SHIFT = (-.030, -.088, -.188)          # lpips.ScalingLayer
LPIPS_SCALE = (.458, .448, .450)


class _LpipsBase(torch.nn.Module):
    def __init__(self, chns, input_range):
        super().__init__()
        self.chns = chns
        self.input_range = input_range       # '255': images in [0,255] (the projector's convention); 'pm1': images in [-1,1] (lpips)
        self.register_buffer('shift', torch.tensor(LPIPS_SHIFT))
        self.register_buffer('scale', torch.tensor(LPIPS_SCALE))
        for i, c in enumerate(chns):
            self.add_module(f'lin{i}', _Lin(c))

    accepts_cl4 = True          # [N,4,H,W] channels-last images with a zero fourth channel (hipops / loss_nets.image_prepare) are taken as they are

    def _input(self, img):
        pre_mul, pre_add = (2.0 / 255.0, -1.0) if self.input_range == '255' else (1.0, 0.0)
        if img.dim() == 4 and img.shape[1] == 4 and H.is_cl(img) and img.dtype == torch.float32:
            # the projector's one-pass image (scale, shift, area resize, 4-float pixels): the LPIPS input normalisation is one fused multiply-add on it
            # (fourth channel: 0 * x + 0) instead of slice / scale / shift / divide / concatenate / layout passes (13 launches + their backward)
            key = (self.scale.data_ptr(), self.scale._version, self.shift.data_ptr(), self.shift._version, img.device)
            if getattr(self, '_cl4_key', None) != key:          # (constants of the network: formed once, not nine tiny launches per replayed step)
                with torch.no_grad():
                    self._cl4_a = torch.cat([pre_mul / self.scale, self.scale.new_zeros(1)]).view(1, 4, 1, 1).to(img.device)
                    self._cl4_b = torch.cat([(pre_add - self.shift) / self.scale, self.scale.new_zeros(1)]).view(1, 4, 1, 1).to(img.device)
                self._cl4_key = key
            return torch.addcmul(self._cl4_b, img, self._cl4_a)
        return _image_cl4(img, self.shift, self.scale, pre_mul, pre_add)

    def _head(self, taps):
        return lpips_features(taps, [getattr(self, f'lin{i}').sqrt_weight() for i in range(len(self.chns))])

    def distance(self, a, b):
        """LPIPS distance per image, [N]."""
        return (self(a) - self(b)).square().sum(1)


class VGG16LPIPS(_LpipsBase):
    """LPIPS-VGG as a feature extractor: forward(img) -> [N, F] with  sum((f(a)-f(b))^2) == LPIPS_vgg(a, b).  Taps: relu1_2, relu2_2,
    relu3_3, relu4_3, relu5_3 (torchvision children 3, 8, 15, 22, 29).  The projector feeds 256^2 images in [0,255]."""
    TAPS = (3, 8, 15, 22, 29)

    def __init__(self, input_range='255', seed: Optional[int] = 11):
        super().__init__((64, 128, 256, 512, 512), input_range)
        self.net = _VGG16Trunk()
        if seed is not None:
            _he_init_(self, seed)

    def forward(self, img):
        _, outs = self.net.run(self._input(img), self.TAPS[-1], self.TAPS)
        return self._head([outs[t] for t in self.TAPS])


class VGG16Features(torch.nn.Module):
    """torchvision vgg16().features children 0..upto as a spatial feature map [N,C,h,w] (upto=14: conv3_3 before its ReLU, the
    reference's layers='14').  Takes the image as the reference passes it (no extra normalisation, warping_loss.py:35-36)."""

    def __init__(self, upto=14, seed: Optional[int] = 12):
        super().__init__()
        self.upto = upto
        self.features = _VGG16Trunk()
        if seed is not None:
            _he_init_(self, seed)

    def forward(self, img):
        n, c, h, w = img.shape
        x = torch.cat([img.float(), img.new_zeros(n, 1, h, w, dtype=torch.float32)], 1).contiguous(memory_format=torch.channels_last)
        y, _ = self.features.run(x, self.upto, ())
        return y


class _AlexTrunk(torch.nn.Module):
    """torchvision.models.alexnet().features children 0..11 split as lpips.pretrained_networks.alexnet does
    (slice1 = 0-1, slice2 = 2-4, slice3 = 5-7, slice4 = 8-9, slice5 = 10-11); parameter keys `slice<k>.<child>.weight`."""
    SPEC = ((1, 0, 3, 64, 11, 4, 2, False), (2, 3, 64, 192, 5, 1, 2, True), (3, 6, 192, 384, 3, 1, 1, True), (4, 8, 384, 256, 3, 1, 1, False),
            (5, 10, 256, 256, 3, 1, 1, False))          # (slice, child index, cin, cout, k, stride, pad, max-pool 3/2 first)

    def __init__(self):
        super().__init__()
        for sl, idx, cin, cout, k, s, p, _ in self.SPEC:
            m = torch.nn.Module()
            conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p)
            conv.requires_grad_(False)
            m.add_module(str(idx), conv)
            self.add_module(f'slice{sl}', m)

    def run(self, x):
        outs = []
        for sl, idx, cin, cout, k, s, p, pool in self.SPEC:
            conv = getattr(getattr(self, f'slice{sl}'), str(idx))
            if pool:
                x = max_pool(x, 3, 2)
            x = conv_act(x, conv.weight, conv.bias, s, p, 'relu')
            outs.append(x)
        return outs


class LPIPSAlex(_LpipsBase):
    """lpips.LPIPS(net='alex') as a feature extractor (images in [-1,1]): sum((f(a)-f(b))^2) == lpips(a, b)."""

    def __init__(self, input_range='pm1', seed: Optional[int] = 13):
        super().__init__((64, 192, 384, 256, 256), input_range)
        self.net = _AlexTrunk()
        if seed is not None:
            _he_init_(self, seed)

    def forward(self, img):
        return self._head(self.net.run(self._input(img)))
