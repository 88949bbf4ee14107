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
        """xyz1 [4, res*res] of the surface points o + d * depth (batch 1)."""
        res = depth.shape[-1]
        if ray_origin.shape[0] == 1 and ray_origin.shape[1] == res ** 2:
            ray_origin = ray_origin.squeeze(0).reshape(res, res, 3).permute(2, 0, 1)
        if ray_dirs.shape[0] == 1 and ray_dirs.shape[1] == res ** 2:
            ray_dirs = ray_dirs.squeeze(0).reshape(res, res, 3).permute(2, 0, 1)
        xyz = ray_origin + ray_dirs * depth.squeeze(0)
        ones = torch.ones(1, res, res, device=xyz.device)
        return torch.cat([xyz, ones], dim=0).reshape(4, res * res)
