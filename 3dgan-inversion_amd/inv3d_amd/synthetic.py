"""Synthetic generator construction + deterministic weights / latents / cameras for benchmarking and smoke tests
(no checkpoints or datasets are reachable offline).  The per-tensor generator is keyed by state-dict name exactly as
SURVEY.md section 8d prescribes, so CPU-oracle and GPU runs see identical inputs on any machine."""
import math
from typing import Dict, Optional, Tuple

import torch

from .training.triplane import TriPlaneGenerator


def default_rendering_kwargs() -> dict:
    return dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                disparity_space_sampling=False, clamp_mode='softplus', white_back=False,
                superresolution_module='training.superresolution.SuperresolutionHybrid8XDC', superresolution_noise_mode='none',
                sr_antialias=True, c_gen_conditioning_zero=False, c_scale=1.0, decoder_lr_mul=1.0, avg_camera_radius=2.7,
                avg_camera_pivot=[0, 0, 0.2], density_reg=0.25, density_reg_p_dist=0.004, reg_type='l1')


def make_generator(w_dim=512, z_dim=512, c_dim=25, plane_res=256, channel_base=32768, channel_max=512, nrr=128, sr_in_res=128,
                   sr_widths=(256, 128), mapping_layers=2, rendering_kwargs: Optional[dict] = None, device='cuda') -> TriPlaneGenerator:
    """ffhqrebalanced512-128-shaped generator by default; smaller sizes for tests."""
    rk = default_rendering_kwargs() if rendering_kwargs is None else dict(rendering_kwargs)
    rk.setdefault('superresolution_module', 'training.superresolution.SuperresolutionHybrid8XDC')
    G = TriPlaneGenerator(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, img_resolution=sr_in_res * 4, img_channels=3, sr_num_fp16_res=4,
                          mapping_kwargs={'num_layers': mapping_layers}, rendering_kwargs=rk,
                          sr_kwargs={'channel_base': channel_base, 'channel_max': channel_max, 'fused_modconv_default': 'inference_only',
                                     'sr_widths': tuple(sr_widths), 'input_resolution': sr_in_res, 'w_dim': w_dim},
                          plane_resolution=plane_res, channel_base=channel_base, channel_max=channel_max,
                          fused_modconv_default='inference_only', conv_clamp=None)
    G.neural_rendering_resolution = nrr
    return G.eval().float().to(device)


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def randn_named(name, seed, shape):
    g = torch.Generator(device='cpu').manual_seed(_name_seed(name, seed))
    return torch.randn(shape, generator=g, dtype=torch.float32)


def rand_named(name, seed, shape):
    g = torch.Generator(device='cpu').manual_seed(_name_seed(name, seed))
    return torch.rand(shape, generator=g, dtype=torch.float32)


@torch.no_grad()
def load_synthetic_weights(G: torch.nn.Module, seed: int = 0, bias_scale: float = 0.1) -> Dict[str, torch.Tensor]:
    """Fill every parameter/buffer of G in place: conv/FC weights and const ~ N(0,1); affine bias = 1; other biases ~
    bias_scale*N(0,1); noise_strength ~ U(0,0.1); noise_const ~ N(0,1); mapping fc weights ~ N(0,1)/0.01; w_avg = 0."""
    f1 = torch.tensor([1., 3., 3., 1.])
    fir = torch.outer(f1, f1)
    fir = fir / fir.sum()
    sd = G.state_dict()
    out = {}
    for name, ref in sd.items():
        shape = tuple(ref.shape)
        if name.endswith('resample_filter'):
            t = fir.clone()
        elif name.endswith('affine.bias'):
            t = torch.ones(shape)
        elif name.endswith('noise_strength'):
            t = rand_named(name, seed, shape) * 0.1
        elif name.endswith('w_avg'):
            t = torch.zeros(shape)
        elif name.endswith('.bias'):
            t = randn_named(name, seed, shape) * bias_scale
        elif name.startswith('backbone.mapping.fc') and name.endswith('.weight'):
            t = randn_named(name, seed, shape) / 0.01
        else:
            t = randn_named(name, seed, shape)
        out[name] = t
    G.load_state_dict({k: v.to(sd[k].device) for k, v in out.items()})
    return out


@torch.no_grad()
def apply_heavy_tail(G: torch.nn.Module, seed: int = 0) -> None:
    """Trained-checkpoint statistics on top of load_synthetic_weights (no EG3D pickle exists in the image): per-input-channel log-normal gains on every
    modulated conv weight (unit mean square), four x100 channels in b4.const, two x10 entries in every affine bias, noise_strength ~ U(0,1).  The f16x3
    operand split normalises each tensor by ONE power of two taken from max|x| * max|style|: this is the input that stresses it (training/
    networks_stylegan2.py:54-56 pre-normalises for the same reason).  Same numbers as oracle.eg3d_oracle.heavy_tailed_params (tests compare the two)."""
    sd = G.state_dict()
    out = {}
    for name, t in sd.items():
        body = name.startswith('backbone.synthesis.') or name.startswith('superresolution.')
        dev = t.device
        if body and name.endswith('.weight') and t.dim() == 4:
            g = torch.exp(randn_named(name + '#gain', seed, (t.shape[1],)))
            g = g / g.square().mean().sqrt()
            out[name] = t * g.to(dev)[None, :, None, None]
        elif name.endswith('b4.const'):
            idx = torch.randperm(t.shape[0], generator=torch.Generator().manual_seed(_name_seed(name + '#outliers', seed)))[:4]
            v = t.clone(); v[idx.to(dev)] *= 100.0
            out[name] = v
        elif body and name.endswith('affine.bias'):
            idx = torch.randperm(t.shape[0], generator=torch.Generator().manual_seed(_name_seed(name + '#outliers', seed)))[:2]
            v = t.clone(); v[idx.to(dev)] *= 10.0
            out[name] = v
        elif name.endswith('noise_strength'):
            out[name] = rand_named(name + '#heavy', seed, tuple(t.shape)).to(dev)
    sd.update(out)
    G.load_state_dict(sd)


def synth_ws(num_ws: int, w_dim: int, n: int, seed: int = 1, wplus: bool = False) -> torch.Tensor:
    if wplus:
        return 0.5 * randn_named('ws+', seed, (n, num_ws, w_dim))
    return (0.5 * randn_named('ws', seed, (n, 1, w_dim))).repeat(1, num_ws, 1)


def lookat_cam2world(origin: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    fwd = torch.nn.functional.normalize(target - origin, dim=0)
    up = torch.tensor([0., 1., 0.])
    right = -torch.nn.functional.normalize(torch.linalg.cross(up, fwd), dim=0)
    up2 = torch.nn.functional.normalize(torch.linalg.cross(fwd, right), dim=0)
    m = torch.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up2, fwd, origin
    return m


def synth_cameras(n: int, seed: int = 2, radius: float = 2.7, focal: float = 4.2647) -> torch.Tensor:
    """c[N,25] = [cam2world(16), intrinsics(9)]: look-at-origin from `radius`, yaw +-0.35, pitch +-0.25 rad around pi/2."""
    ang = rand_named('cams', seed, (n, 2))
    yaw = math.pi / 2 + (ang[:, 0] * 2 - 1) * 0.35
    pitch = math.pi / 2 + (ang[:, 1] * 2 - 1) * 0.25
    cams = []
    for i in range(n):
        h, v = float(yaw[i]), float(pitch[i])
        origin = torch.tensor([radius * math.sin(v) * math.cos(math.pi - h), radius * math.cos(v),
                               radius * math.sin(v) * math.sin(math.pi - h)], dtype=torch.float32)
        cams.append(lookat_cam2world(origin, torch.zeros(3)))
    c2w = torch.stack(cams)
    K = torch.tensor([focal, 0, 0.5, 0, focal, 0.5, 0, 0, 1], dtype=torch.float32)
    return torch.cat([c2w.reshape(n, 16), K[None].repeat(n, 1)], 1)


def make_uniforms(n: int, rays: int, dc: int, df: int, seed: int = 4) -> Tuple[torch.Tensor, torch.Tensor]:
    return rand_named('u1', seed, (n, rays, dc, 1)), rand_named('u2', seed, (n * rays, df))
