"""RaySampler with the reference's interface (training/volumetric_rendering/ray_sampler.py:24-93): camera matrices ->
per-pixel ray origins/directions, differentiable w.r.t. cam2world and intrinsics (pose optimisation), on gfx950."""
import torch

from ... import fused


class RaySampler(torch.nn.Module):
    def forward(self, cam2world_matrix, intrinsics, resolution, need_cam_space=False):
        if need_cam_space:
            # the camera-space variant (ray_sampler.py:66-71; no caller in the repository): camera origin, normalised pinhole directions
            # and the pixel-centre grid -- a handful of tensor ops, differentiable w.r.t. the intrinsics
            n, r = cam2world_matrix.shape[0], int(resolution)
            dev = cam2world_matrix.device
            fx, fy, cx, cy, sk = (intrinsics[:, 0, 0], intrinsics[:, 1, 1], intrinsics[:, 0, 2], intrinsics[:, 1, 2], intrinsics[:, 0, 1])
            ar = (torch.arange(r, dtype=torch.float32, device=dev) + 0.5) / r
            uv = torch.stack(torch.meshgrid(ar, ar, indexing='ij')).flip(0).reshape(2, -1).transpose(1, 0).unsqueeze(0).repeat(n, 1, 1)
            x_cam, y_cam = uv[:, :, 0], uv[:, :, 1]
            x_lift = (x_cam - cx[:, None] + cy[:, None] * sk[:, None] / fy[:, None] - sk[:, None] * y_cam / fy[:, None]) / fx[:, None]
            y_lift = (y_cam - cy[:, None]) / fy[:, None]
            dirs_cam = torch.nn.functional.normalize(torch.stack((x_lift, y_lift, torch.ones_like(x_lift)), dim=-1), dim=2)
            return torch.zeros_like(cam2world_matrix[:, :3, 3]), dirs_cam, uv
        return fused.RayGenFn.apply(cam2world_matrix, intrinsics, int(resolution))

    def calculate_xyz_of_depth(self, ray_origin, ray_dirs, depth):
        """Surface points of one view, homogeneous: origins / directions [1,res^2,3] (or [3,res,res]) and depth [1,1,res,res] ->
        [4, res^2] = (o + d * depth ; 1), pixel-major (reference: ray_sampler.py:75-93; the warping loss's caller, warping_loss.py:20)."""
        res = depth.shape[-1]
        flat = lambda v: v.reshape(res * res, 3) if v.shape[-1] == 3 else v.reshape(3, res * res).t()       # noqa: E731
        xyz = flat(ray_origin) + flat(ray_dirs) * depth.reshape(res * res, 1)
        return torch.cat([xyz, xyz.new_ones(res * res, 1)], dim=1).t()
