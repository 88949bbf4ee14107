"""StyleGAN2 generator modules backed by the gfx950 kernels.

Module tree, constructor arguments and state-dict keys follow the reference's generator half
(training/networks_stylegan2.py: FullyConnectedLayer :95, MappingNetwork :190, SynthesisLayer :272, ToRGBLayer :338,
SynthesisBlock :362, SynthesisNetwork :468, Generator :525) so that checkpoints and callers are interchangeable; the
bodies are different: every layer is one fused autograd op (inv3d_amd.fused) working on channels_last fp32 tensors.

Numerics: always fp32 (>= the reference: its fp16 SR path, :421-424, is computed in fp32 here), conv_clamp kept.
`fused_modconv` is accepted and ignored: the activation-scaled formulation used here is the reference's non-fused branch
(:70-79), interchangeable to ~2e-6."""
import math

import numpy as np
import torch

from .. import fused
from .. import hipops as H
from ..reference_binding import ReferenceStateMixin
from ..torch_utils.ops import bias_act, upfirdn2d


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True, flip_weight=True,
                     fused_modconv=True):
    """The reference's stand-alone operator (networks_stylegan2.py:34-91) with its signature: style-modulated, optionally demodulated
    convolution with FIR-filtered resampling and an additive noise tensor.  Executed in the activation-scaled form (the reference's
    `fused_modconv=False` branch, :70-79 -- numerically interchangeable with the grouped-conv branch; `fused_modconv` is accepted and
    ignored): x * styles -> conv2d_resample -> * demodulation coefficients (+ noise).  The generator's layers do NOT go through this
    function (they are single fused ops, fused.ModConvLayerFn); it exists for callers of the operator itself."""
    from ..torch_utils.ops import conv2d_resample
    n = x.shape[0]
    co, ci, kh, kw = weight.shape
    if tuple(styles.shape) != (n, ci) or x.shape[1] != ci:
        raise ValueError(f'modulated_conv2d: x {tuple(x.shape)}, weight {tuple(weight.shape)}, styles {tuple(styles.shape)} do not agree')
    dcoefs = None
    if demodulate:          # rsqrt(sum_{i,taps} (w * s)^2 + 1e-8) = rsqrt(sum_i s_i^2 sum_taps w^2 + 1e-8)   (:62-65)
        dcoefs = torch.rsqrt(styles.square() @ weight.square().sum((2, 3)).t() + 1e-8)
    y = conv2d_resample.conv2d_resample(x=x * styles.to(x.dtype).reshape(n, -1, 1, 1), w=weight.to(x.dtype), f=resample_filter, up=up, down=down,
                                        padding=padding, flip_weight=flip_weight)
    if dcoefs is not None:
        y = y * dcoefs.to(x.dtype).reshape(n, -1, 1, 1)
    if noise is not None:
        y = y + noise.to(x.dtype)
    return y


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """x / rms(x) along `dim` (the latent / embedding normalisation in front of the mapping MLP)."""
    return x * torch.rsqrt(torch.mean(x * x, dim=dim, keepdim=True) + eps)


class FullyConnectedLayer(ReferenceStateMixin, torch.nn.Module):
    """y = act(x @ (W * g_w)^T + b * g_b) with the equalised-learning-rate gains g_w = lr_mul / sqrt(in), g_b = lr_mul
    (reference: networks_stylegan2.py:95-127; state-dict keys `weight` [out,in], `bias` [out])."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features, self.out_features, self.activation = int(in_features), int(out_features), activation
        self.weight_gain, self.bias_gain = lr_multiplier / math.sqrt(in_features), lr_multiplier
        self.weight = torch.nn.Parameter(torch.randn(out_features, in_features) * (1.0 / lr_multiplier))
        self.bias = torch.nn.Parameter(torch.full((out_features,), float(bias_init), dtype=torch.float32)) if bias else None

    def forward(self, x):
        # off the per-layer hot path (the 26 style affines and the decoder run inside fused kernels): one small library GEMM
        bias = None if self.bias is None else self.bias.to(x.dtype) * self.bias_gain
        y = torch.nn.functional.linear(x, self.weight.to(x.dtype) * self.weight_gain, bias)
        return y if self.activation == 'linear' else bias_act.bias_act(y, None, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


class MappingNetwork(ReferenceStateMixin, torch.nn.Module):
    """z (+ embedded c) -> w, broadcast to num_ws rows, truncated towards the running mean `w_avg`
    (reference: networks_stylegan2.py:190-268; children `embed`, `fc0..fc{L-1}`, buffer `w_avg`)."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None, activation='lrelu',
                 lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        embed = 0 if c_dim == 0 else (embed_features if embed_features is not None else w_dim)
        hidden = layer_features if layer_features is not None else w_dim
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed)
        widths = [z_dim + embed] + [hidden] * (num_layers - 1) + [w_dim]
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            self.add_module(f'fc{i}', FullyConnectedLayer(fan_in, fan_out, activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros(w_dim))

    def _truncate(self, w, psi):
        return self.w_avg + (w - self.w_avg) * psi                      # lerp(w_avg, w, psi)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        parts = []
        if self.z_dim > 0:
            parts.append(normalize_2nd_moment(z.float()))
        if self.c_dim > 0:
            parts.append(normalize_2nd_moment(self.embed(c.float())))
        w = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        for i in range(self.num_layers):
            w = self._modules[f'fc{i}'](w)
        if update_emas and self.w_avg_beta is not None:                  # exponential moving average of the batch mean
            self.w_avg.copy_(self.w_avg * self.w_avg_beta + w.detach().mean(0) * (1 - self.w_avg_beta))
        if self.num_ws is not None:
            w = w[:, None, :].expand(-1, self.num_ws, -1).contiguous()
        if truncation_psi == 1:
            return w
        if self.num_ws is None or truncation_cutoff is None:
            return self._truncate(w, truncation_psi)
        head = self._truncate(w[:, :truncation_cutoff], truncation_psi)
        return torch.cat([head, w[:, truncation_cutoff:]], dim=1)


class SynthesisLayer(ReferenceStateMixin, torch.nn.Module):
    """Modulated 3x3 conv (optionally up-2) + noise + bias + lrelu as ONE fused op (reference: networks_stylegan2.py:272-335).
    State: weight [O,I,3,3], bias [O], affine.{weight,bias} (bias initialised to 1), noise_strength [], buffers noise_const [r,r] and
    resample_filter [4,4]."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        if activation != 'lrelu' or kernel_size != 3 or up not in (1, 2) or list(resample_filter) != [1, 3, 3, 1]:
            raise NotImplementedError('the fused layer op covers the EG3D generator configuration: 3x3, lrelu, up in {1,2}, [1,3,3,1] FIR')
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.padding = 1
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels, 3, 3))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        if use_noise:
            self.noise_strength = torch.nn.Parameter(torch.zeros(()))
            self.register_buffer('noise_const', torch.randn(resolution, resolution))
        self._cache = fused.WeightCache()

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, noise_inject=None, styles=None, demod=None, single_consumer=False,
                input_is_layer_output=False, precision=None, next_styles=None, defer_epilogue=False, rgb_head=None):
        """`rgb_head`: (toRGB layer, its styles) when that layer reads the returned tensor next (fused.ModConvLayerFn).
        `defer_epilogue`: the returned tensor goes to a toRGB node next and to nothing before it (fused.ModConvLayerFn).
        `styles` / `demod`: this layer's affine(w) and demodulation coefficients when the enclosing network already evaluated them
        for all layers in one launch (fused.style_bank).  `single_consumer`: the caller promises that the returned tensor feeds exactly one
        fused op (conv1 / toRGB of the same block), which lets that op's backward absorb this layer's activation backward (fused.py);
        `input_is_layer_output`: x is such a layer's output, handed over directly."""
        assert noise_mode in ['random', 'const', 'none']
        if styles is None:
            styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode != 'none' and noise_inject is not None:
            noise = noise_inject             # per-sample noise [N,1,res,res] supplied by the caller: recorded draws ('random') or the
        elif self.use_noise and noise_mode == 'random':      # per-image optimised noise maps of a batched latent projection ('const')
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device)
        elif self.use_noise and noise_mode == 'const':
            noise = self.noise_const
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return fused.ModConvLayerFn.apply(x, self.weight, styles, noise, self.noise_strength if noise is not None else None, self.bias,
                                          self.up, self.act_gain * gain, clamp, self._cache, self.weight.requires_grad, demod, single_consumer,
                                          input_is_layer_output, precision, next_styles, defer_epilogue,
                                          None if rgb_head is None else (rgb_head[0].weight, rgb_head[1], rgb_head[0].bias, rgb_head[0].conv_clamp, rgb_head[0]._cache))

    def extra_repr(self):
        return f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}, ' \
               f'resolution={self.resolution:d}, up={self.up}, activation={self.activation:s}'


class ToRGBLayer(ReferenceStateMixin, torch.nn.Module):
    """Modulated 1x1 conv without demodulation + bias, accumulated onto the skip image (reference: networks_stylegan2.py:338-359)."""

    out_pad = True          # fused.ToRGBFn computes ceil(out_channels / 4) * 4 channels: zero-padded weight images (fused.prepack_weights)

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        if kernel_size != 1:
            raise NotImplementedError('toRGB layers are 1x1')
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.weight_gain = in_channels ** -0.5                 # folded into the styles (networks_stylegan2.py:354)
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels, 1, 1))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self._cache = fused.WeightCache()

    def forward(self, x, w, fused_modconv=True, skip=None, styles=None, passthrough=False, input_is_layer_output=False, skip_up=False):
        """Returns skip + torgb(x) on a channel count padded to a multiple of 4 (padding channels stay as in `skip`/zero).
        `styles`: affine(w) * weight_gain when precomputed by the enclosing network.  passthrough=True: returns (img, x) where the
        second output is x routed through this op's autograd node (see fused.ToRGBFn)."""
        if styles is None:
            styles = self.affine(w) * self.weight_gain
        return fused.ToRGBFn.apply(x, self.weight, styles, self.bias, skip, self.conv_clamp, self._cache, self.weight.requires_grad, passthrough,
                                   input_is_layer_output, skip_up)


def _pad4(c):
    return (c + 3) // 4 * 4


class SynthesisBlock(ReferenceStateMixin, torch.nn.Module):
    """One resolution of the 'skip' architecture: [conv0 (up-2)] -> conv1 -> toRGB accumulated onto the up-sampled skip image; the 4^2
    block starts from the learned constant (reference: networks_stylegan2.py:362-465)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False, fused_modconv_default=True,
                 conv0_up=2, **layer_kwargs):
        if architecture != 'skip':
            raise NotImplementedError("EG3D generators use the 'skip' architecture")
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels, self.is_last = in_channels, w_dim, resolution, img_channels, is_last
        self.architecture, self.use_fp16, self.fused_modconv_default = architecture, use_fp16, fused_modconv_default
        self.conv0_up = int(conv0_up)           # 1: SynthesisBlockNoUp (training/superresolution.py:155-262) -- conv0 keeps the resolution, the skip image is added as it is
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        first = in_channels == 0
        common = dict(w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        if first:
            self.const = torch.nn.Parameter(torch.randn(out_channels, resolution, resolution))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, up=self.conv0_up, resample_filter=resample_filter, **common)
        self.conv1 = SynthesisLayer(out_channels, out_channels, **common)
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_conv, self.num_torgb = (1 if first else 2), 1

    def packed_layers(self):
        """The layers of this block that keep packed weight images (for fused.prepack_weights)."""
        return ([] if self.in_channels == 0 else [self.conv0]) + [self.conv1, self.torgb]

    def affine_entries(self, w_idx):
        """(affine module, ws row, post scale) of this block's modulated layers in evaluation order (for fused.style_bank)."""
        ent = []
        if self.in_channels != 0:
            ent.append((self.conv0.affine, w_idx + len(ent), 1.0, self.conv0))
        ent.append((self.conv1.affine, w_idx + len(ent), 1.0, self.conv1))
        ent.append((self.torgb.affine, w_idx + len(ent), self.torgb.weight_gain, None))
        return ent

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, noise_inject=None, _name='', styles=None,
                **layer_kwargs):
        """x: [N,Cin,r/2,r/2] (any layout) or None; img: skip image with channels padded to a multiple of 4, or None.
        `styles`: (styles, demods) of this block's layers in the order of affine_entries(), or None.  Returns (x, img) as channels_last fp32."""
        w_iter = iter(ws.unbind(dim=1))
        s_iter = iter(styles[0]) if styles is not None else iter(lambda: None, 0)
        d_iter = iter(styles[1]) if styles is not None else iter(lambda: None, 0)
        ni = noise_inject or {}
        if self.use_fp16 and not force_fp32:        # the reference runs this block in fp16 (networks_stylegan2.py:421-424): one product of fp16-rounded operands
            layer_kwargs = dict(layer_kwargs, precision='f16x1')
        if self.in_channels == 0:
            if self.const.is_cuda and not self.const.requires_grad:      # frozen (latent projection): the broadcast channels_last copy is made once
                nb = int(ws.shape[0])
                x = H.memo(('const_cl', nb), [self.const], lambda: (lambda t: H.tag_amax(t, H.absmax(t)))(H.to_cl(self.const.detach().float().unsqueeze(0).expand(nb, -1, -1, -1))))
            else:
                x = self.const.unsqueeze(0).expand(ws.shape[0], -1, -1, -1)
            x = self.conv1(x, next(w_iter), noise_inject=ni.get(f'{_name}.conv1'), styles=next(s_iter), demod=next(d_iter), single_consumer=True,
                           defer_epilogue=True, **layer_kwargs)
        else:       # conv0's output feeds conv1 only, conv1's the toRGB node only (which passes it on to the next block through itself)
            s0, s1 = next(s_iter), next(s_iter)          # conv1's styles are known before conv0 runs: its epilogue can write conv1's operand image
            st = next(s_iter)                             # ... and the toRGB layer's before conv1 runs: a <= 4-channel toRGB can ride in conv1's epilogue
            s_iter = iter([st])
            x = self.conv0(x, next(w_iter), noise_inject=ni.get(f'{_name}.conv0'), styles=s0, demod=next(d_iter), single_consumer=True,
                           next_styles=s1, **layer_kwargs)
            x = self.conv1(x, next(w_iter), noise_inject=ni.get(f'{_name}.conv1'), styles=s1, demod=next(d_iter), single_consumer=True,
                           input_is_layer_output=True, defer_epilogue=True,
                           rgb_head=(self.torgb, st) if (st is not None and self.torgb.out_channels <= 4) else None, **layer_kwargs)
        # the skip image is handed over at half resolution: the toRGB node up-samples it (inside its conv's epilogue where it can)
        skip_up = img is not None and self.conv0_up == 2
        if self.is_last:
            img = self.torgb(x, next(w_iter), skip=img, styles=next(s_iter), input_is_layer_output=True, skip_up=skip_up)
        else:       # x goes on to the next block: route it through the toRGB node so the two gradients are summed in its epilogue
            amax = getattr(x, '_eg3d_amax', None)
            img, x = self.torgb(x, next(w_iter), skip=img, styles=next(s_iter), passthrough=True, input_is_layer_output=True, skip_up=skip_up)
            if amax is not None:        # the pass-through output is the same values: keep the producer's max|x| report with it
                H.tag_amax(x, amax)
            x._eg3d_from_torgb = True       # (the next block's conv0 may leave the finish of its split-K data gradient to this node's backward launch)
        return x, img


class SynthesisBlockNoUp(SynthesisBlock):
    """The first block of the 128^2 / 256^2 super-resolution heads (reference: training/superresolution.py:155-262): conv0 at the block's own
    resolution, skip image added without up-sampling."""

    def __init__(self, *args, **kw):
        kw['conv0_up'] = 1
        super().__init__(*args, **kw)


class SynthesisNetwork(ReferenceStateMixin, torch.nn.Module):
    """Blocks b4 ... b{img_resolution}; width(res) = min(channel_base / res, channel_max); num_ws = number of conv layers + the last
    block's toRGB (reference: networks_stylegan2.py:468-522)."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        if img_resolution < 4 or img_resolution & (img_resolution - 1):
            raise ValueError('img_resolution must be a power of two >= 4')
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels, self.num_fp16_res = w_dim, img_resolution, img_channels, num_fp16_res
        self.img_resolution_log2 = img_resolution.bit_length() - 1
        self.block_resolutions = [4 << i for i in range(self.img_resolution_log2 - 1)]
        width = lambda r: min(channel_base // r, channel_max)               # noqa: E731
        self.num_ws = 0
        for res in self.block_resolutions:
            last = res == img_resolution
            block = SynthesisBlock(0 if res == 4 else width(res // 2), width(res), w_dim=w_dim, resolution=res, img_channels=img_channels,
                                   is_last=last, use_fp16=False, **block_kwargs)
            self.add_module(f'b{res}', block)
            self.num_ws += block.num_conv + (block.num_torgb if last else 0)

    def _draw_noise(self, n, device, prefix):
        """noise_mode='random': the fresh per-layer normal draws of one forward (networks_stylegan2.py:318-319) from ONE generator launch,
        handed to the layers as views (13 launches of a few microseconds each otherwise)."""
        lay = []
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            for name in (('conv1',) if block.in_channels == 0 else ('conv0', 'conv1')):
                if getattr(block, name).use_noise:
                    lay.append((f'{prefix}.b{res}.{name}', res))
        if not lay:
            return None
        flat = torch.randn(sum(n * r * r for _, r in lay), device=device)
        out, o = {}, 0
        for key, r in lay:
            out[key] = flat[o:o + n * r * r].view(n, 1, r, r)
            o += n * r * r
        return out

    def forward(self, ws, noise_inject=None, _prefix='backbone.synthesis', _extra_entries=None, _bank_out=None, _extra_pack=None, **block_kwargs):
        """_extra_entries / _bank_out: style-bank entries of a network that runs later on the same ws (the super-resolution head) ride along in
        this network's bank launch -- and in its backward launches; their (styles, demods) are left in _bank_out['extra'].  _extra_pack: that
        network's weight-carrying layers, re-packed in this network's batched launch when the weights train (their demodulation sums are bank inputs)."""
        ws = ws.to(torch.float32)
        x = img = None
        # all 20 style affines of the backbone in one launch (None: some affine is trainable -> per-layer path)
        entries, counts, w_idx = [], [], 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            ent = block.affine_entries(w_idx)
            entries += ent
            counts.append(len(ent))
            w_idx += block.num_conv
        if ws.is_cuda and self.b4.conv1.weight.requires_grad:       # pivotal tuning: every weight image is stale once per step -> one launch for all
            fused.prepack_weights([m for res in self.block_resolutions for m in getattr(self, f'b{res}').packed_layers()] + list(_extra_pack or []))
        n_own = len(entries)
        extra = list(_extra_entries) if (_extra_entries and _bank_out is not None and n_own + len(_extra_entries) <= fused.L.STYLE_BANK_MAX) else []
        bank = fused.style_bank(ws, entries + extra)
        if bank is not None and extra:
            _bank_out['extra'] = (bank[0][n_own:], bank[1][n_own:])
            bank = (bank[0][:n_own], bank[1][:n_own])
        elif bank is None and extra:          # the joint bank does not apply (a non-linear affine somewhere): each network on its own
            bank = fused.style_bank(ws, entries)
        if noise_inject is None and block_kwargs.get('noise_mode', 'random') == 'random':
            noise_inject = self._draw_noise(ws.shape[0], ws.device, _prefix)
        w_idx = s_idx = 0
        for res, cnt in zip(self.block_resolutions, counts):
            block = getattr(self, f'b{res}')
            cur = ws.narrow(1, w_idx, block.num_conv + block.num_torgb)
            w_idx += block.num_conv
            st = (bank[0][s_idx:s_idx + cnt], bank[1][s_idx:s_idx + cnt]) if bank is not None else None
            s_idx += cnt
            x, img = block(x, img, cur, noise_inject=noise_inject, _name=f'{_prefix}.b{res}', styles=st, **block_kwargs)
        return img if img.shape[1] == self.img_channels else img[:, :self.img_channels]


class Generator(ReferenceStateMixin, torch.nn.Module):
    """mapping + synthesis (reference: networks_stylegan2.py:525-553); EG3D uses it as the tri-plane backbone."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
