"""RaySampler with the reference's interface (training/volumetric_rendering/ray_sampler.py:24-93): camera matrices ->
per-pixel ray origins/directions, differentiable w.r.t. cam2world and intrinsics (pose optimisation), on gfx950."""
import torch

from ... import fused


class RaySampler(torch.nn.Module):
    def forward(self, cam2world_matrix, intrinsics, resolution, need_cam_space=False):
        if need_cam_space:
            raise NotImplementedError('need_cam_space=True is not used on the inversion path')
        return fused.RayGenFn.apply(cam2world_matrix, intrinsics, int(resolution))

    def calculate_xyz_of_depth(self, ray_origin, ray_dirs, depth):
        """Surface points of one view, homogeneous: origins / directions [1,res^2,3] (or [3,res,res]) and depth [1,1,res,res] ->
        [4, res^2] = (o + d * depth ; 1), pixel-major (reference: ray_sampler.py:75-93; the warping loss's caller, warping_loss.py:20)."""
        res = depth.shape[-1]
        flat = lambda v: v.reshape(res * res, 3) if v.shape[-1] == 3 else v.reshape(3, res * res).t()       # noqa: E731
        xyz = flat(ray_origin) + flat(ray_dirs) * depth.reshape(res * res, 1)
        return torch.cat([xyz, xyz.new_ones(res * res, 1)], dim=1).t()
