"""TriPlaneGenerator / OSGDecoder with the reference's interface (training/triplane.py:18-136):
    G.synthesis(ws[N,14,512], c[N,25], neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                use_cached_backbone=False, **synthesis_kwargs{noise_mode, force_fp32, fused_modconv})
        -> {'image' [N,3,512,512], 'image_raw' [N,3,128,128], 'image_depth' [N,1,128,128]}
Same attribute tree / state-dict keys (backbone.{synthesis,mapping}, superresolution.block{0,1}, decoder.net.{0,2}), so
scripts/run_pti.py-style callers, the projector and the coaches can use it unchanged.  Extra, optional synthesis kwargs for
deterministic runs: render_uniforms=(u1,u2), noise_inject={layer-name: [N,1,res,res]}."""
import torch

from .networks_stylegan2 import Generator as StyleGAN2Backbone, FullyConnectedLayer
from .superresolution import SuperresolutionHybrid8XDC
from .volumetric_rendering.ray_sampler import RaySampler
from .volumetric_rendering.renderer import ImportanceRenderer

_SR_MODULES = {'training.superresolution.SuperresolutionHybrid8XDC': SuperresolutionHybrid8XDC}


class OSGDecoder(torch.nn.Module):
    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        lr = options['decoder_lr_mul']
        self.net = torch.nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim, lr_multiplier=lr), torch.nn.Softplus(),
                                       FullyConnectedLayer(self.hidden_dim, 1 + options['decoder_output_dim'], lr_multiplier=lr))

    def forward(self, sampled_features, ray_directions):
        """Stand-alone decode of pre-sampled features [N,3,M,C] (API parity; the hot path decodes inside the render kernel)."""
        x = sampled_features.mean(1)
        N, M, C = x.shape
        x = self.net(x.reshape(N * M, C)).view(N, M, -1)
        return {'rgb': torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001, 'sigma': x[..., 0:1]}


class TriPlaneGenerator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, sr_num_fp16_res=0, mapping_kwargs={}, rendering_kwargs={},
                 sr_kwargs={}, plane_resolution=256, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.renderer = ImportanceRenderer()
        self.ray_sampler = RaySampler()
        synthesis_kwargs.pop('num_fp16_res', None)
        self.backbone = StyleGAN2Backbone(z_dim, c_dim, w_dim, img_resolution=plane_resolution, img_channels=32 * 3,
                                          mapping_kwargs=mapping_kwargs, num_fp16_res=0, **synthesis_kwargs)
        sr_cls = _SR_MODULES.get(rendering_kwargs.get('superresolution_module', 'training.superresolution.SuperresolutionHybrid8XDC'))
        if sr_cls is None:
            raise NotImplementedError(f"superresolution module {rendering_kwargs.get('superresolution_module')} (only the 512^2 head is on the inversion path)")
        sr_kwargs = dict(sr_kwargs)          # the SR blocks take 512-d latents unless told otherwise (superresolution.py:268-275 hard-codes w_dim=512)
        self.superresolution = sr_cls(channels=32, img_resolution=img_resolution, sr_num_fp16_res=sr_num_fp16_res,
                                      sr_antialias=rendering_kwargs.get('sr_antialias', True), **sr_kwargs)
        self.decoder = OSGDecoder(32, {'decoder_lr_mul': rendering_kwargs.get('decoder_lr_mul', 1), 'decoder_output_dim': 32})
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self._last_planes = None

    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        if self.rendering_kwargs['c_gen_conditioning_zero']:
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0), truncation_psi=truncation_psi,
                                     truncation_cutoff=truncation_cutoff, update_emas=update_emas)

    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
                  render_uniforms=None, noise_inject=None, **synthesis_kwargs):
        cam2world = c[:, :16].view(-1, 4, 4)
        intrinsics = c[:, 16:25].view(-1, 3, 3)
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        synthesis_kwargs.pop('force_fp32', None)          # always fp32 here
        ray_origins, ray_directions = self.ray_sampler(cam2world, intrinsics, neural_rendering_resolution)
        N = ray_origins.shape[0]
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            planes = self.backbone.synthesis(ws, update_emas=update_emas, noise_inject=noise_inject, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        if render_uniforms is not None:
            self.renderer.set_uniforms(*render_uniforms)
        feat, depth, _ = self.renderer(planes, self.decoder, ray_origins, ray_directions, self.rendering_kwargs)
        Hh = Ww = self.neural_rendering_resolution
        # [N, H*W, 32] is already the channels_last image of [N,32,H,W]
        feature_image = feat.view(N, Hh, Ww, feat.shape[-1]).permute(0, 3, 1, 2)
        depth_image = depth.permute(0, 2, 1).reshape(N, 1, Hh, Ww)
        rgb_image = feature_image[:, :3].contiguous()
        sr_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'noise_mode'}
        sr_image = self.superresolution(rgb_image, feature_image, ws, noise_mode=self.rendering_kwargs['superresolution_noise_mode'],
                                        noise_inject=noise_inject, **sr_kwargs)
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image}

    def sample(self, coordinates, directions, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.sample_mixed(coordinates, directions, ws, update_emas=update_emas, **synthesis_kwargs)

    def sample_mixed(self, coordinates, directions, ws, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        synthesis_kwargs.pop('force_fp32', None)
        planes = self.backbone.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
        return self.renderer.run_model(planes, self.decoder, coordinates, directions, self.rendering_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)
