"""TriPlaneGenerator / OSGDecoder with the reference's interface (training/triplane.py:18-136):
    G.synthesis(ws[N,14,512], c[N,25], neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                use_cached_backbone=False, **synthesis_kwargs{noise_mode, force_fp32, fused_modconv})
        -> {'image' [N,3,512,512], 'image_raw' [N,3,128,128], 'image_depth' [N,1,128,128]}
Same attribute tree / state-dict keys (backbone.{synthesis,mapping}, superresolution.block{0,1}, decoder.net.{0,2}), so
scripts/run_pti.py-style callers, the projector and the coaches can use it unchanged.  Extra, optional synthesis kwargs for
deterministic runs: render_uniforms=(u1,u2), noise_inject={layer-name: [N,1,res,res]}."""
import os

import torch

from .. import fused
from .. import hipops as H
from .. import graphed
from ..reference_binding import ReferenceStateMixin
from .networks_stylegan2 import Generator as StyleGAN2Backbone, FullyConnectedLayer
from .superresolution import (SuperresolutionHybrid8XDC, SuperresolutionHybrid8X, SuperresolutionHybrid4X, SuperresolutionHybrid2X,
                              SuperresolutionHybridDeepfp32)
from .volumetric_rendering.ray_sampler import RaySampler
from .volumetric_rendering.renderer import ImportanceRenderer

JOINT_STYLE_BANK = os.environ.get('EG3D_JOINT_STYLE_BANK', '1') != '0'   # SR head's style affines computed by the backbone's bank launch
SHARE_FEATURE_NODE = True      # image_raw's 4-float copy and the SR head's input behind one autograd node (fused.slice_rgb4 share=True): their gradients summed in one launch

_SR_MODULES = {'training.superresolution.SuperresolutionHybrid8XDC': SuperresolutionHybrid8XDC,
               'training.superresolution.SuperresolutionHybrid8X': SuperresolutionHybrid8X,
               'training.superresolution.SuperresolutionHybrid4X': SuperresolutionHybrid4X,
               'training.superresolution.SuperresolutionHybrid2X': SuperresolutionHybrid2X,
               'training.superresolution.SuperresolutionHybridDeepfp32': SuperresolutionHybridDeepfp32}


class OSGDecoder(ReferenceStateMixin, torch.nn.Module):
    """Tri-plane feature decoder 32 -> 64 -(softplus)-> 1 + 32 (reference: training/triplane.py:116-136; state `net.0`, `net.2`).
    On the hot path these four tensors are consumed inside the render kernels; forward() is the stand-alone form."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        fc = lambda i, o: FullyConnectedLayer(i, o, lr_multiplier=options['decoder_lr_mul'])        # noqa: E731
        self.net = torch.nn.Sequential(fc(n_features, self.hidden_dim), torch.nn.Softplus(), fc(self.hidden_dim, 1 + options['decoder_output_dim']))

    def forward(self, sampled_features, ray_directions):
        """[N,3,M,C] plane features -> {'rgb' [N,M,32] in (-0.001, 1.001), 'sigma' [N,M,1]} (mean over the three planes first)."""
        feats = sampled_features.mean(dim=1)
        out = self.net(feats.flatten(0, 1)).unflatten(0, feats.shape[:2])
        sigma, colour = out[..., :1], out[..., 1:]
        return {'rgb': torch.sigmoid(colour) * 1.002 - 0.001, 'sigma': sigma}


class TriPlaneGenerator(ReferenceStateMixin, torch.nn.Module):
    """backbone (StyleGAN2 -> 3 x 32 x 256^2 planes) -> renderer (ray sampling + two-pass volume rendering with the OSG decoder) ->
    super-resolution head.  Constructor arguments, children and state-dict keys as the reference class (training/triplane.py:18-46)."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={}, rendering_kwargs={},
                 sr_kwargs={}, plane_resolution=256, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.rendering_kwargs = rendering_kwargs
        self.neural_rendering_resolution = 64
        self._last_planes = None
        sr_name = rendering_kwargs.get('superresolution_module', 'training.superresolution.SuperresolutionHybrid8XDC')
        if sr_name not in _SR_MODULES:
            raise NotImplementedError(f'superresolution module {sr_name}: not one of the reference\'s five heads (training/superresolution.py)')
        synthesis_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'num_fp16_res'}          # the backbone runs in fp32 (:40)
        self.renderer = ImportanceRenderer()
        self.ray_sampler = RaySampler()
        self.backbone = StyleGAN2Backbone(z_dim, c_dim, w_dim, img_resolution=plane_resolution, img_channels=96, mapping_kwargs=mapping_kwargs,
                                          num_fp16_res=0, **synthesis_kwargs)
        self.superresolution = _SR_MODULES[sr_name](channels=32, img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res,
                                                    sr_antialias=rendering_kwargs.get('sr_antialias', True), **dict(sr_kwargs))
        self.decoder = OSGDecoder(32, {'decoder_lr_mul': rendering_kwargs.get('decoder_lr_mul', 1), 'decoder_output_dim': 32})

    # ---- latent side ---------------------------------------------------------------------------------------------------------
    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        """Camera-conditioned mapping: c is scaled by rendering_kwargs['c_scale'] (zeroed with c_gen_conditioning_zero) (triplane.py:48-51)."""
        rk = self.rendering_kwargs
        cond = torch.zeros_like(c) if rk['c_gen_conditioning_zero'] else c
        return self.backbone.mapping(z, cond * rk.get('c_scale', 0), truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff,
                                     update_emas=update_emas)

    def _planes(self, ws, use_cached, cache, update_emas, noise_inject, kwargs):
        if use_cached and self._last_planes is not None:
            planes = self._last_planes
        else:
            with H._Span('backbone_fwd'):
                planes = self.backbone.synthesis(ws, update_emas=update_emas, noise_inject=noise_inject, **kwargs)
            H.span_between_grads('backbone_bwd', planes, ws)
        if cache:
            self._last_planes = planes
        return planes

    # ---- image side ----------------------------------------------------------------------------------------------------------
    sr_fp16_default = False      # True: G.synthesis without force_fp32 runs the fp16 blocks like the reference does on CUDA (see `sr_fp16`)

    graph_eager = True           # plain calls whose signature repeats are replayed from HIP graphs (inv3d_amd/graphed.py); False: always per-launch

    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
                  render_uniforms=None, noise_inject=None, sr_fp16=None, **synthesis_kwargs):
        """ws [N,num_ws,w_dim], c [N,25] = (cam2world 4x4, intrinsics 3x3) -> {'image','image_raw','image_depth'} (triplane.py:53-90).
        Arithmetic: fp32-equivalent everywhere by default.  `sr_fp16=True` (and no `force_fp32=True`) runs the blocks the reference runs
        in fp16 when `force_fp32` is not passed -- the super-resolution head, sr_num_fp16_res > 0, networks_stylegan2.py:421-424; what
        BaseCoach.forward does during pivotal tuning -- with one product of fp16-rounded operands (EG3D_PREC_F16X1; accumulation and
        storage stay fp32).  `render_uniforms=(u1,u2)` / `noise_inject` pin the stratified-sampling / per-layer noise draws.
        A call signature that repeats (an optimisation loop) is captured into HIP graphs and replayed -- same kernels, same results, one
        launch per direction instead of ~190 (inv3d_amd/graphed.py; `graph_eager`)."""
        kw = dict(synthesis_kwargs)
        for name, val, default in (('neural_rendering_resolution', neural_rendering_resolution, None), ('update_emas', update_emas, False),
                                   ('cache_backbone', cache_backbone, False), ('use_cached_backbone', use_cached_backbone, False),
                                   ('noise_inject', noise_inject, None), ('sr_fp16', sr_fp16 if sr_fp16 is not None else self.sr_fp16_default, False)):
            if val is not default and val != default:
                kw[name] = val
        return graphed.synthesis(self, self._synthesis_impl, ws, c, render_uniforms=render_uniforms, **kw)

    def _synthesis_impl(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
                        render_uniforms=None, noise_inject=None, sr_fp16=None, **synthesis_kwargs):
        """The per-launch synthesis (what `synthesis` captures)."""
        if sr_fp16 is None:
            sr_fp16 = self.sr_fp16_default
        block_fp32 = bool(synthesis_kwargs.get('force_fp32', False)) or not sr_fp16
        if neural_rendering_resolution is not None:
            self.neural_rendering_resolution = neural_rendering_resolution                           # sticky, as in the reference (:58-61)
        res = self.neural_rendering_resolution
        kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'force_fp32'}
        origins, directions = self.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), res)
        # the SR head's six style affines ride along in the backbone's bank launch (and its two backward launches) when the backbone runs
        bank_out, bkw = {}, kwargs
        sr_entries = getattr(self.superresolution, 'bank_entries', None)
        if JOINT_STYLE_BANK and sr_entries is not None and ws.is_cuda and ws.dtype == torch.float32 and not (use_cached_backbone and self._last_planes is not None):
            bkw = dict(kwargs, _extra_entries=sr_entries(ws), _bank_out=bank_out, _extra_pack=self.superresolution.block0.packed_layers() + self.superresolution.block1.packed_layers())
        planes = self._planes(ws, use_cached_backbone, cache_backbone, update_emas, noise_inject, bkw)
        if render_uniforms is not None:
            self.renderer.set_uniforms(*render_uniforms)
        feat, depth, _ = self.renderer(planes, self.decoder, origins, directions, self.rendering_kwargs)
        n = origins.shape[0]
        if feat.is_cuda and feat.shape[-1] % 4 == 0:      # 4-float pixels (channel 3 = 0) for the SR head's skip path and the fused loss kernels
            raw4 = fused.slice_rgb4(feat, res, share=SHARE_FEATURE_NODE)
            if SHARE_FEATURE_NODE:                        # the SR head reads the features behind the same node: one gradient pass for both
                raw4, feat = raw4
            rgb = raw4[:, :3]                             # image_raw is a view of it
            rgb._eg3d_padded4 = raw4
        features = feat.view(n, res, res, feat.shape[-1]).permute(0, 3, 1, 2)          # [N, H*W, 32] IS the channels_last image: zero-copy
        if not (feat.is_cuda and feat.shape[-1] % 4 == 0):
            rgb = features[:, :3].contiguous()
        image = self.superresolution(rgb, features, ws, noise_mode=self.rendering_kwargs['superresolution_noise_mode'], noise_inject=noise_inject,
                                     force_fp32=block_fp32, **({'_bank': bank_out['extra']} if 'extra' in bank_out else {}),
                                     **{k: v for k, v in kwargs.items() if k != 'noise_mode'})
        return {'image': image, 'image_raw': rgb, 'image_depth': depth.transpose(1, 2).reshape(n, 1, res, res)}

    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        """{'rgb','sigma'} of the field at arbitrary points for given ws (triplane.py:99-103; density-grid extraction)."""
        kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'force_fp32'}
        planes = self._planes(ws, False, False, update_emas, None, kwargs)
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def sample(self, coordinates, directions, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        """sample_mixed on mapped latents (triplane.py:92-97)."""
        return self.sample_mixed(coordinates, directions, self.mapping(z, c, truncation_psi, truncation_cutoff, update_emas),
                                 update_emas=update_emas, **synthesis_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        """synthesis on mapped latents (triplane.py:105-108)."""
        return self.synthesis(self.mapping(z, c, truncation_psi, truncation_cutoff, update_emas), c, update_emas=update_emas,
                              neural_rendering_resolution=neural_rendering_resolution, cache_backbone=cache_backbone,
                              use_cached_backbone=use_cached_backbone, **synthesis_kwargs)
