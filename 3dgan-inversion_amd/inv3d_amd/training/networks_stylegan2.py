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
from ..torch_utils.ops import bias_act, upfirdn2d


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class FullyConnectedLayer(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())          # tiny library GEMM ([N,512] x [512,C])
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None, activation='lrelu',
                 lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        embed_features = w_dim if embed_features is None else embed_features
        if c_dim == 0:
            embed_features = 0
        layer_features = w_dim if layer_features is None else layer_features
        feats = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for i in range(num_layers):
            setattr(self, f'fc{i}', FullyConnectedLayer(feats[i], feats[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        x = None
        if self.z_dim > 0:
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        if update_emas and self.w_avg_beta is not None:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        assert activation == 'lrelu' and kernel_size == 3 and up in (1, 2), 'fused layer op covers the EG3D generator configuration'
        assert list(resample_filter) == [1, 3, 3, 1]
        self.in_channels, self.out_channels, self.w_dim, self.resolution, self.up = in_channels, out_channels, w_dim, resolution, up
        self.use_noise, self.activation, self.conv_clamp = use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self._cache = fused.WeightCache()

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, noise_inject=None, styles=None, demod=None):
        """`styles` / `demod`: this layer's affine(w) and demodulation coefficients when the enclosing network already evaluated them
        for all layers in one launch (fused.style_bank)."""
        assert noise_mode in ['random', 'const', 'none']
        if styles is None:
            styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = noise_inject if noise_inject is not None else \
                torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device)
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return fused.ModConvLayerFn.apply(x, self.weight, styles, noise, self.noise_strength if noise is not None else None, self.bias,
                                          self.up, self.act_gain * gain, clamp, self._cache, self.weight.requires_grad, demod)

    def extra_repr(self):
        return f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}, ' \
               f'resolution={self.resolution:d}, up={self.up}, activation={self.activation:s}'


class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        assert kernel_size == 1
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / math.sqrt(in_channels * (kernel_size ** 2))
        self._cache = fused.WeightCache()

    def forward(self, x, w, fused_modconv=True, skip=None, styles=None, passthrough=False):
        """Returns skip + torgb(x) on a channel count padded to a multiple of 4 (padding channels stay as in `skip`/zero).
        `styles`: affine(w) * weight_gain when precomputed by the enclosing network.  passthrough=True: returns (img, x) where the
        second output is x routed through this op's autograd node (see fused.ToRGBFn)."""
        if styles is None:
            styles = self.affine(w) * self.weight_gain
        return fused.ToRGBFn.apply(x, self.weight, styles, self.bias, skip, self.conv_clamp, self._cache, self.weight.requires_grad, passthrough)


def _pad4(c):
    return (c + 3) // 4 * 4


class SynthesisBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False, fused_modconv_default=True,
                 **layer_kwargs):
        assert architecture == 'skip', "EG3D generators use the 'skip' architecture"
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels, self.is_last = in_channels, w_dim, resolution, img_channels, is_last
        self.architecture, self.use_fp16, self.fused_modconv_default = architecture, use_fp16, fused_modconv_default
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        if in_channels != 0:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, **layer_kwargs)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
        self.num_torgb += 1

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
        if self.in_channels == 0:
            x = self.const.unsqueeze(0).expand(ws.shape[0], -1, -1, -1)
            x = self.conv1(x, next(w_iter), noise_inject=ni.get(f'{_name}.conv1'), styles=next(s_iter), demod=next(d_iter), **layer_kwargs)
        else:
            x = self.conv0(x, next(w_iter), noise_inject=ni.get(f'{_name}.conv0'), styles=next(s_iter), demod=next(d_iter), **layer_kwargs)
            x = self.conv1(x, next(w_iter), noise_inject=ni.get(f'{_name}.conv1'), styles=next(s_iter), demod=next(d_iter), **layer_kwargs)
        if img is not None:
            img = fused.UpsampleImgFn.apply(img)
        if self.is_last:
            img = self.torgb(x, next(w_iter), skip=img, styles=next(s_iter))
        else:       # x goes on to the next block: route it through the toRGB node so the two gradients are summed in its epilogue
            img, x = self.torgb(x, next(w_iter), skip=img, styles=next(s_iter), passthrough=True)
        return x, img


class SynthesisNetwork(torch.nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels, self.num_fp16_res = w_dim, img_resolution, img_channels, num_fp16_res
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.num_ws = 0
        for res in self.block_resolutions:
            cin = channels[res // 2] if res > 4 else 0
            block = SynthesisBlock(cin, channels[res], w_dim=w_dim, resolution=res, img_channels=img_channels,
                                   is_last=(res == img_resolution), use_fp16=False, **block_kwargs)
            self.num_ws += block.num_conv
            if res == img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

    def forward(self, ws, noise_inject=None, _prefix='backbone.synthesis', **block_kwargs):
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
        bank = fused.style_bank(ws, entries)
        w_idx = s_idx = 0
        for res, cnt in zip(self.block_resolutions, counts):
            block = getattr(self, f'b{res}')
            cur = ws.narrow(1, w_idx, block.num_conv + block.num_torgb)
            w_idx += block.num_conv
            st = (bank[0][s_idx:s_idx + cnt], bank[1][s_idx:s_idx + cnt]) if bank is not None else None
            s_idx += cnt
            x, img = block(x, img, cur, noise_inject=noise_inject, _name=f'{_prefix}.b{res}', styles=st, **block_kwargs)
        return img if img.shape[1] == self.img_channels else img[:, :self.img_channels]


class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
