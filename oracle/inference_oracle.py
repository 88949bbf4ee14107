"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

Restatement of the inference consumers of the hot path: the orbit cameras of the interpolation video,
the mean-latent statistics of the projector and the density grid of the shape extraction.  Pinned against the imported reference by
tests/golden/make_golden.py::gen_inference (fixture tests/golden/inference.npz): LookAtPoseSampler and create_samples are the
reference's own functions run here; the density grid is the reference renderer's run_model evaluated the way create_geometry does.
"""
import math

import numpy as np
import torch

from . import eg3d_oracle as O


def lookat_pose(h, v, lookat, radius):
    """LookAtPoseSampler.sample with zero stddev (utils/camera_utils.py:87-105) + create_cam2world_matrix (:137-156).  h, v floats."""
    v = min(max(float(v), 1e-5), math.pi - 1e-5)
    theta = torch.tensor(float(h))
    phi = torch.arccos(torch.tensor(1 - 2 * (v / math.pi)))
    origin = torch.stack([radius * torch.sin(phi) * torch.cos(math.pi - theta), radius * torch.cos(phi),
                          radius * torch.sin(phi) * torch.sin(math.pi - theta)]).float()
    fwd = torch.nn.functional.normalize(torch.as_tensor(lookat, dtype=torch.float32) - origin, dim=0)
    up = torch.tensor([0., 1., 0.])
    right = -torch.nn.functional.normalize(torch.linalg.cross(up, fwd), dim=0)
    up = torch.nn.functional.normalize(torch.linalg.cross(fwd, right), dim=0)
    m = torch.eye(4)
    m[:3, :3] = torch.stack((right, up, fwd), -1)
    m[:3, 3] = origin
    return m


def orbit_cameras(num_frames=240, yaw_range=0.35, pitch_range=0.25, radius=2.7, focal=4.2647, lookat=(0., 0., 0.)):
    """Camera of every frame of gen_interp_video (gen_videos.py:105-117; the reference writes 3.14 for pi) -> [F,25]."""
    K = torch.tensor([focal, 0, 0.5, 0, focal, 0.5, 0, 0, 1.])
    cams = []
    for i in range(num_frames):
        m = lookat_pose(3.14 / 2 + yaw_range * np.sin(2 * 3.14 * i / num_frames), 3.14 / 2 - 0.05 + pitch_range * np.cos(2 * 3.14 * i / num_frames),
                        lookat, radius)
        cams.append(torch.cat([m.reshape(16), K]))
    return torch.stack(cams)


def create_samples(N, cube_length):
    """Grid points of the shape extraction (training/coaches/single_id_coach.py:165-186, voxel_origin = 0): point i has
    (x, y, z) = (i // N^2, (i // N) % N, i % N) * voxel_size - cube_length / 2, computed through the reference's float divisions."""
    origin = -cube_length / 2
    vs = cube_length / (N - 1)
    idx = torch.arange(0, N ** 3, dtype=torch.long)
    s = torch.zeros(N ** 3, 3)
    s[:, 2] = idx % N
    s[:, 1] = (idx.float() / N) % N
    s[:, 0] = ((idx.float() / N) / N) % N
    return (s * vs + origin).unsqueeze(0)


def density_grid(P, cfg, ws, res, pad=None, pad_value=-1000.0):
    """create_geometry (single_id_coach.py:120-157): sigma of run_model at create_samples(res, box_warp), reshaped [res]^3, flipped along
    axis 0, border of `pad` voxels set to pad_value."""
    planes = O.backbone_synthesis(P, cfg, ws, noise_mode='const')
    planes = planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])
    pts = create_samples(res, cfg.rendering['box_warp'])
    sigma = O.run_model(P, planes, pts, cfg.rendering)[1]
    g = torch.flip(sigma.reshape(res, res, res), [0])
    pad = int(30 * res / 256) if pad is None else pad
    if pad > 0:
        g = g.clone()
        g[:pad] = pad_value; g[-pad:] = pad_value
        g[:, :pad] = pad_value; g[:, -pad:] = pad_value
        g[:, :, :pad] = pad_value; g[:, :, -pad:] = pad_value
    return g


def w_stats(P, cfg, num_samples, seed=123, psi=0.7, cutoff=14):
    """Mean latent and its spread (training/projectors/w_projector.py:88-97): z ~ RandomState(123), canonical camera as conditioning."""
    cam = torch.cat([lookat_pose(math.pi / 2, math.pi / 2, (0., 0., 0.), 2.7).reshape(1, 16),
                     torch.tensor([[4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1.]])], 1)
    z = torch.from_numpy(np.random.RandomState(seed).randn(num_samples, cfg.z_dim)).float()
    w = O.mapping(P, cfg, z, cam.repeat(num_samples, 1), psi, cutoff)[:, :1, :].numpy().astype(np.float32)
    w_avg = np.mean(w, axis=0, keepdims=True)
    w_std = (np.sum((w - w_avg) ** 2) / num_samples) ** 0.5
    return torch.from_numpy(w_avg), float(w_std)
