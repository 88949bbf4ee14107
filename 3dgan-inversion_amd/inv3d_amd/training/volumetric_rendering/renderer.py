"""ImportanceRenderer with the reference's interface (training/volumetric_rendering/renderer.py:137-203), executed by the
fused per-ray gfx950 kernel: forward(planes, decoder, ray_origins, ray_directions, rendering_options) ->
(rgb [N,M,32], depth [N,M,1], weights_sum [N,M,1]);  run_model(planes, decoder, coords, dirs, options) -> {'rgb','sigma'}.

Ray marching (reference ray_marcher.py:25-57, MipRayMarcher2) happens inside the render kernels; there is no stand-alone marcher module.

Randomness: the stratified jitter / importance uniforms are drawn with torch.rand on the device unless the caller
injects them (`set_uniforms`, used by parity tests and by deterministic optimisation runs)."""
import torch

from ... import fused
from ... import hipops as H
from . import math_utils


def generate_planes():
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


def _planes_cl(planes):
    """[N,3,C,H,W] (or [N,3C,H,W]) -> channels_last [N,3C,H,W] view/copy."""
    if planes.dim() == 5:
        n, p, c, h, w = planes.shape
        planes = planes.reshape(n, p * c, h, w)
    return H.to_cl(planes.float())


class ImportanceRenderer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.plane_axes = generate_planes()
        self._u = None

    def set_uniforms(self, u1, u2):
        """Inject the uniforms of the next forward: u1 [N,M,Dc,1] (stratified), u2 [N*M,Df] (importance)."""
        self._u = (u1, u2)

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        opts = rendering_options
        N, M, _ = ray_origins.shape
        Dc, Df = int(opts['depth_resolution']), int(opts['depth_resolution_importance'])
        dev = ray_origins.device
        if self._u is not None:
            u1, u2 = self._u
            self._u = None
        else:
            u1 = torch.rand((N, M, Dc, 1), device=dev)
            u2 = torch.rand((N * M, Df), device=dev) if Df > 0 else None
        limits = None
        if opts['ray_start'] == opts['ray_end'] == 'auto':
            rs, re = math_utils.get_ray_limits_box(ray_origins, ray_directions, box_side_length=opts['box_warp'])
            ok = re > rs
            lo = torch.where(ok, rs, torch.full_like(rs, float('inf'))).min()
            hi = torch.where(ok, rs, torch.full_like(rs, float('-inf'))).max()
            any_ok = ok.any()
            rs = torch.where(ok | ~any_ok, rs, lo)
            re = torch.where(ok | ~any_ok, re, hi)
            limits = torch.cat([rs, re], -1)
        net = decoder.net
        rgb, depth, wsum = fused.RenderFn.apply(_planes_cl(planes), ray_origins, ray_directions, net[0].weight, net[0].bias, net[2].weight,
                                                net[2].bias, u1, u2, opts, float(opts.get('decoder_lr_mul', 1)), limits)
        return rgb, depth, wsum

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        """Density/colour query at arbitrary points (no gradients; used for shape extraction)."""
        import math
        pl = _planes_cl(planes)
        net = decoder.net
        lr = float(options.get('decoder_lr_mul', 1))
        hid, cin = net[0].weight.shape
        with torch.no_grad():
            w0 = (net[0].weight.float() * (lr / math.sqrt(cin))).contiguous()
            b0 = (net[0].bias.float() * lr).contiguous()
            w1t = (net[2].weight.float() * (lr / math.sqrt(hid))).t().contiguous()
            b1 = (net[2].bias.float() * lr).contiguous()
            coords = sample_coordinates.contiguous().float()
            N, M, _ = coords.shape
            dummy = torch.zeros(1, device=coords.device)
            p = H.make_render_params(pl, dummy.expand(1, 1, 1), dummy.expand(1, 1, 1), dummy, None,
                                     dict(options, depth_resolution=2, depth_resolution_importance=0,
                                          ray_start=0.0 if options['ray_start'] == 'auto' else options['ray_start'],
                                          ray_end=1.0 if options['ray_end'] == 'auto' else options['ray_end']),
                                     w0, b0, w1t, b1, None, None, None, None, None)
            rgb, sigma = H.sample_decode(p, coords, M)
        if options.get('density_noise', 0) > 0:
            sigma = sigma + torch.randn_like(sigma) * options['density_noise']
        return {'rgb': rgb, 'sigma': sigma}
