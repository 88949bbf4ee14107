"""Ray/box helpers used by the 'auto' ray-limit mode (reference: training/volumetric_rendering/math_utils.py:46-118).
Tiny per-ray arithmetic kept as device-side tensor glue; the per-ray limits are then fed to the fused renderer."""
import torch


def normalize_vecs(vectors):
    return vectors / torch.norm(vectors, dim=-1, keepdim=True)


def torch_dot(x, y):
    return (x * y).sum(-1)


def get_ray_limits_box(rays_o, rays_d, box_side_length):
    """Slab test against the axis-aligned cube of side `box_side_length`; misses are flagged (-1, -2)."""
    shp = rays_o.shape
    o = rays_o.detach().reshape(-1, 3)
    d = rays_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    inv = 1 / d
    lo = (-half - o) * inv
    hi = (half - o) * inv
    near = torch.minimum(lo, hi)
    far = torch.maximum(lo, hi)
    # sequential axis folding, as the slab algorithm does (x,y then z)
    tmin, tmax = near[:, 0], far[:, 0]
    valid = ~((tmin > far[:, 1]) | (near[:, 1] > tmax))
    tmin = torch.maximum(tmin, near[:, 1])
    tmax = torch.minimum(tmax, far[:, 1])
    valid &= ~((tmin > far[:, 2]) | (near[:, 2] > tmax))
    tmin = torch.maximum(tmin, near[:, 2])
    tmax = torch.minimum(tmax, far[:, 2])
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2))
    return tmin.reshape(*shp[:-1], 1), tmax.reshape(*shp[:-1], 1)


def linspace(start, stop, num):
    steps = torch.arange(num, dtype=torch.float32, device=start.device) / (num - 1)
    steps = steps.reshape([-1] + [1] * start.ndim)
    return start[None] + steps * (stop - start)[None]
