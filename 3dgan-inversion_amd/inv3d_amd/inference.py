"""Inference consumers of the hot path: everything here is a no-grad caller of G.synthesis /
G.mapping / ImportanceRenderer.run_model on the gfx950 kernels.

  lookat_pose, orbit_cameras   <- utils/camera_utils.py:87-105,137-156 and the per-frame cameras of gen_videos.py:105-117
  render_orbit                 <- gen_interp_video, gen_videos.py:63-146 (one latent, 240-frame orbit; no video encoder here: frames are
                                  returned as tensors)
  estimate_w_stats             <- mean latent / spread of the projector, training/projectors/w_projector.py:88-97
  density_grid                 <- create_geometry + create_samples, training/coaches/single_id_coach.py:120-186 (sigma on a res^3 grid)

Differences from the reference that do not change results: the tri-planes of a fixed latent are synthesised once per orbit / per grid
instead of once per frame / per chunk (the reference re-runs the backbone every time), and grid coordinates are generated per chunk on
the device instead of materialising res^3 x 3 floats on the host."""
import math
from typing import Iterator, Optional, Tuple

import numpy as np
import torch


def lookat_pose(h: float, v: float, lookat=(0., 0., 0.), radius: float = 2.7, device='cpu') -> torch.Tensor:
    """cam2world [4,4] of a camera on the sphere of `radius` at yaw h, pitch v looking at `lookat` (y up, no roll)."""
    v = min(max(float(v), 1e-5), math.pi - 1e-5)
    theta = torch.tensor(float(h))
    phi = torch.arccos(torch.tensor(1 - 2 * (v / math.pi)))
    origin = torch.stack([radius * torch.sin(phi) * torch.cos(math.pi - theta), radius * torch.cos(phi),
                          radius * torch.sin(phi) * torch.sin(math.pi - theta)]).float()
    fwd = torch.nn.functional.normalize(torch.as_tensor(lookat, dtype=torch.float32) - origin, dim=0)
    up0 = torch.tensor([0., 1., 0.])
    right = -torch.nn.functional.normalize(torch.linalg.cross(up0, fwd), dim=0)
    up = torch.nn.functional.normalize(torch.linalg.cross(fwd, right), dim=0)
    m = torch.eye(4)
    m[:3, :3] = torch.stack((right, up, fwd), -1)
    m[:3, 3] = origin
    return m.to(device)


def orbit_cameras(num_frames: int = 240, yaw_range: float = 0.35, pitch_range: float = 0.25, radius: float = 2.7, focal: float = 4.2647,
                  lookat=(0., 0., 0.), device='cpu') -> torch.Tensor:
    """[F,25] conditioning vectors (cam2world 16 + intrinsics 9) of the reference's orbit; it spells pi as 3.14 and so does this."""
    K = torch.tensor([focal, 0, 0.5, 0, focal, 0.5, 0, 0, 1.])
    cams = []
    for i in range(num_frames):
        m = lookat_pose(3.14 / 2 + yaw_range * np.sin(2 * 3.14 * i / num_frames), 3.14 / 2 - 0.05 + pitch_range * np.cos(2 * 3.14 * i / num_frames),
                        lookat, radius)
        cams.append(torch.cat([m.reshape(16), K]))
    return torch.stack(cams).to(device)


@torch.no_grad()
def render_orbit(G, ws: torch.Tensor, num_frames: int = 240, image_mode: str = 'image', cameras: Optional[torch.Tensor] = None,
                 **synthesis_kwargs) -> Iterator[torch.Tensor]:
    """Yields one [3,H,W] (or [1,h,w] for image_depth, normalised to [-1,1] as gen_videos.py:143-145) frame per camera for the single
    latent ws [1,num_ws,w_dim].  The backbone runs once; every frame is rendering + super-resolution."""
    dev = ws.device
    cams = orbit_cameras(num_frames, device=dev) if cameras is None else cameras.to(dev)
    kw = dict(noise_mode='const', **synthesis_kwargs)
    for i in range(cams.shape[0]):
        out = G.synthesis(ws[:1], cams[i:i + 1], cache_backbone=(i == 0), use_cached_backbone=(i > 0), **kw)
        img = out[image_mode][0]
        if image_mode == 'image_depth':
            img = -img
            img = (img - img.min()) / (img.max() - img.min()) * 2 - 1
        yield img


@torch.no_grad()
def estimate_w_stats(G, num_samples: int = 10000, seed: int = 123, truncation_psi: float = 0.7, truncation_cutoff: int = 14,
                     batch: int = 2048) -> Tuple[torch.Tensor, float]:
    """(w_avg [1,1,w_dim], w_std): statistics of the first latent row over `num_samples` mapped z ~ N(0,1) (numpy RandomState(seed), as
    the reference) conditioned on the canonical frontal camera."""
    dev = next(G.parameters()).device
    cam = torch.cat([lookat_pose(math.pi / 2, math.pi / 2, (0., 0., 0.), 2.7).reshape(1, 16),
                     torch.tensor([[4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1.]])], 1).to(dev)
    z = torch.from_numpy(np.random.RandomState(seed).randn(num_samples, G.z_dim)).float()
    rows = []
    for i in range(0, num_samples, batch):
        zb = z[i:i + batch].to(dev)
        rows.append(G.mapping(zb, cam.expand(zb.shape[0], -1), truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)[:, :1, :].float())
    w = torch.cat(rows, 0)
    w_avg = w.mean(0, keepdim=True)
    w_std = float(((w - w_avg).double().square().sum() / num_samples).sqrt())
    return w_avg, w_std


def _grid_points(res: int, cube_length: float, start: int, stop: int, device) -> torch.Tensor:
    """Points start..stop of create_samples(res, [0,0,0], cube_length): index i -> (i // res^2, (i // res) % res, i % res) scaled to
    [-L/2, L/2].  The reference derives the two slow indices with float32 divisions and `% N`, which leaves fractional parts in the
    coordinates ((i / N) % N is not an integer); reproduced exactly."""
    vs = cube_length / (res - 1)
    idx = torch.arange(start, stop, dtype=torch.long, device=device)
    f = idx.float()
    x = ((f / res) / res) % res
    y = (f / res) % res
    z = (idx % res).float()
    return torch.stack([x, y, z], -1) * vs - cube_length / 2


@torch.no_grad()
def density_grid(G, ws: torch.Tensor, res: int = 512, max_batch: int = 1 << 22, pad: Optional[int] = None, pad_value: float = -1000.0,
                 **synthesis_kwargs) -> torch.Tensor:
    """sigma on the res^3 grid spanning the rendering box, laid out as create_geometry hands it to marching cubes: [res,res,res]
    flipped along axis 0, with a border of int(30*res/256) voxels set to -1000."""
    dev = ws.device
    box = float(G.rendering_kwargs['box_warp'])
    planes = G.backbone.synthesis(ws[:1], noise_mode='const', **synthesis_kwargs)
    planes = planes.view(1, 3, -1, planes.shape[-2], planes.shape[-1])
    total = res ** 3
    sig = torch.empty(total, device=dev)
    for head in range(0, total, max_batch):
        stop = min(total, head + max_batch)
        pts = _grid_points(res, box, head, stop, dev).unsqueeze(0)
        sig[head:stop] = G.renderer.run_model(planes, G.decoder, pts, None, G.rendering_kwargs)['sigma'].reshape(-1)
    g = torch.flip(sig.view(res, res, res), [0]).contiguous()
    pad = int(30 * res / 256) if pad is None else pad
    if pad > 0:
        g[:pad] = pad_value
        g[-pad:] = pad_value
        g[:, :pad] = pad_value
        g[:, -pad:] = pad_value
        g[:, :, :pad] = pad_value
        g[:, :, -pad:] = pad_value
    return g
