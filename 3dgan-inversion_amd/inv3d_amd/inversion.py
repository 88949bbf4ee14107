"""Inversion inner loops on the MI355X path -- the callers of G.synthesis.

  LatentProjector  <- training/projectors/w_projector.py:28-280 (Phase A: latent w [+ camera pose + translation] + noise
                      buffers, LPIPS-feature distance + noise regulariser [+ depth-reprojection warping loss])
  PivotalTuner     <- training/coaches/{base_coach.py:96-126, single_id_coach.py:64-77} (Phase B: all generator weights, Adam 3e-4,
                      MSE(512^2)+MSE(128^2)+LPIPS(512^2)+LPIPS(128^2)+TV(depth), early exit on the LPIPS threshold)

The perceptual networks are third-party weights that are not available offline (VGG16-LPIPS,
torchvision VGG16, LPIPS-AlexNet).  They enter through `feature_net` callables; the default `StubFeatureNet` is a small
fixed-random conv pyramid running on this package's own conv kernels, so the loss has the same structure (feature-space
squared distance) and the same gradient path into the image.  Hyper-parameters default to configs/hyperparameters.py.
All schedule arithmetic stays on the host; there is no per-step device->host sync unless `early_stop` is requested.
"""
import math
import functools
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import hipops
from . import fused as _fused
from .torch_utils.ops import bias_act, conv2d_gradfix


class StubFeatureNet(torch.nn.Module):
    """Stand-in for the LPIPS / VGG feature extractors: 3 x (3x3 conv -> lrelu -> 2x2 avg-pool), fixed random weights,
    unit-normalised channel features at 3 scales, concatenated.  Runs on the gfx950 conv + bias_act kernels."""

    accepts_cl4 = True          # takes [N,4,H,W] channels_last input (channel 3 = 0) as well as RGB

    def __init__(self, widths=(16, 32, 64), seed=1234):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        cin = 4                                  # rgb padded to 4 channels (16-byte pixels)
        self.ws = torch.nn.ParameterList()
        for w in widths:
            self.ws.append(torch.nn.Parameter(torch.randn(w, cin, 3, 3, generator=g) / math.sqrt(cin * 9), requires_grad=False))
            cin = w

    def forward(self, img):
        """Three launches per stage and direction: conv + lrelu (fused epilogue), the 2x2 average, the channel normalisation written
        straight into the flat feature vector (loss_nets.unit_features).  The features are pixel-major inside a stage's slice -- the
        projector only ever takes squared distances between two such vectors."""
        from . import loss_nets as LN
        x = self.stages(img)
        return LN.unit_features(x, eps=1e-10)

    def stages(self, img, upto=None):
        """The pooled stage outputs [N,C_l,h_l,w_l] (channels_last)."""
        from . import loss_nets as LN
        n, c, h, w = img.shape
        x = torch.cat([img, img.new_zeros(n, 4 - c, h, w)], 1) if c < 4 else img
        x = hipops.to_cl(x.float())
        wl = list(self.ws)[:upto]
        if LN.stub_pyramid_ok(x, wl):        # levels this small run as direct convolutions with pooling / activation backward fused (csrc/loss_ops.hip)
            return LN.stub_pyramid(x, wl, 0.2, math.sqrt(2.0))
        outs = []
        for wt in wl:
            x = LN.conv_act(x, wt, None, 1, 1, 'lrelu', 0.2, math.sqrt(2.0))
            x = F.avg_pool2d(x, 2)
            outs.append(x)
        return outs


class _NoiseRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, *bufs):
        reg, grads = hipops.noise_regularizer([b.detach() for b in bufs], scale=scale, want_grad=True)
        ctx.save_for_backward(*grads)
        return reg

    @staticmethod
    def backward(ctx, g):
        grads = ctx.saved_tensors
        return (None,) + tuple(torch._foreach_mul(list(grads), g))


def noise_regularizer(noise_bufs, weight: float = 1.0) -> torch.Tensor:
    """weight * multi-scale shifted auto-correlation penalty on the noise maps (w_projector.py:221-237); value and gradient of
    all buffers come from ONE fused launch (csrc/noise_ops.hip)."""
    return _NoiseRegFn.apply(float(weight), *noise_bufs)


def compute_tv_norm(values: torch.Tensor) -> torch.Tensor:
    """Squared forward-difference total variation of a [*,H,W] map (base_coach.py:294-305)."""
    v00, v01, v10 = values[:, :-1, :-1], values[:, :-1, 1:], values[:, 1:, :-1]
    return ((v00 - v01) ** 2 + (v00 - v10) ** 2).mean()


def quaternion_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """[B,4] (w,x,y,z) -> [B,3,3] (utils/camera_utils.py:201-228)."""
    q = q / torch.sqrt(torch.clamp((q * q).sum(1, keepdim=True), min=1e-8))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    r = torch.stack([1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                     2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                     2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y], 1)
    return r.view(-1, 3, 3)


def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """[B,6] -> [B,3,3]: the continuous 6-D rotation representation of the AFHQ pose head (utils/camera_utils.py:259-273; every
    component is offset by 1e-4 first, as there).  Columns = orthonormalised first vector, second vector, their cross product."""
    v = x.reshape(-1, 2, 3) + 1e-4
    e1 = F.normalize(v[:, 0], dim=-1)
    e2 = F.normalize(v[:, 1] - (e1 * v[:, 1]).sum(-1, keepdim=True) * e1, dim=-1)
    return torch.stack((e1, e2, torch.linalg.cross(e1, e2, dim=-1)), dim=-1)


def euler_to_rotmat(theta: torch.Tensor, phi: torch.Tensor, roll: Optional[torch.Tensor] = None, radius: float = 2.7) -> torch.Tensor:
    """Azimuth / polar angle (+ roll) -> [B,3,3]: rotation of a camera on the sphere looking at the origin, y up -- the rotation
    block of euler2rot -> create_cam2world_matrix_roll (utils/camera_utils.py:241-257, 158-188)."""
    theta, phi = theta.reshape(-1, 1), phi.reshape(-1, 1)
    sp = torch.sin(phi)
    origin = radius * torch.cat([sp * torch.cos(math.pi - theta), torch.cos(phi), sp * torch.sin(math.pi - theta)], 1)
    fwd = F.normalize(-origin, dim=-1, eps=0.0)
    right = -F.normalize(torch.linalg.cross(torch.tensor([0., 1., 0.], device=fwd.device).expand_as(fwd), fwd, dim=-1), dim=-1, eps=0.0)
    up = F.normalize(torch.linalg.cross(fwd, right, dim=-1), dim=-1, eps=0.0)
    rot = torch.stack((right, up, fwd), dim=-1)
    if roll is not None:
        r = roll.reshape(-1, 1).to(rot.device)
        c, s_, z, o = torch.cos(r), torch.sin(r), torch.zeros_like(r), torch.ones_like(r)
        rot = torch.bmm(torch.stack([torch.cat([c, -s_, z], 1), torch.cat([s_, c, z], 1), torch.cat([z, z, o], 1)], 1), rot)
    return rot


POSE_DIMS = {'quat': 4, '6d': 6, 'euler': 2}
POSE_INIT = {'quat': [0., 1., 0., 0.], '6d': [1., 0., 0., 0., -1., 0.], 'euler': [0., 0.]}     # each = the canonical extrinsic's rotation


def pose_to_rotmat(pred: torch.Tensor, mode: str) -> torch.Tensor:
    """The pose-head dispatch of w_projector.py:147-158 / scripts/run_pti.py:36-45: 'quat' (FFHQ, global_config.use_quaternions),
    '6d' (AFHQ, use_6d), 'euler' (two angles added to pi/2, no roll)."""
    if mode == 'quat':
        return quaternion_to_rotmat(pred)
    if mode == '6d':
        return rot6d_to_rotmat(pred)
    if mode == 'euler':
        return euler_to_rotmat(math.pi / 2 + pred[:, 0], math.pi / 2 + pred[:, 1])
    raise ValueError(f'pose_mode must be one of {sorted(POSE_DIMS)}, got {mode!r}')


def pose_to_cam(rotmat: torch.Tensor, translation_opt: torch.Tensor, intrinsic: torch.Tensor, radius: float = 2.7):
    """Rotation + optimisable translation -> extrinsic [B,4,4] and c [B,25] (w_projector.py:160-172)."""
    b = rotmat.shape[0]
    pred_translation = -radius * rotmat[:, :3, 2]
    t_world = -torch.bmm(rotmat, translation_opt.unsqueeze(-1)).squeeze(-1) * radius
    t = t_world + pred_translation
    t = t / torch.norm(t, dim=-1, keepdim=True) * radius
    ext = torch.eye(4, device=rotmat.device).unsqueeze(0).repeat(b, 1, 1)
    ext = torch.cat([torch.cat([rotmat, t.unsqueeze(-1)], 2), ext[:, 3:]], 1)
    return ext, torch.cat([ext.reshape(b, 16), intrinsic.reshape(1, 9).expand(b, 9)], 1)


class _PoseChainFn(torch.autograd.Function):
    """pose vector + translation -> (extrinsic [B,4,4], c [B,25]) in one launch per direction (eg3d_pose_chain_fwd / _bwd: forward-mode duals
    give the 12 x 9 Jacobian with the values).  As PyTorch ops the chain is ~150 one-element kernels forward and as many backward."""

    @staticmethod
    def forward(ctx, pose, translation, intrinsic, radius, mode):
        from . import _lib as L
        pose, translation = pose.contiguous().float(), translation.contiguous().float()
        b = pose.shape[0]
        m = {'quat': 0, '6d': 1, 'euler': 2}[mode]
        dev = pose.device
        ext, cam = torch.empty((b, 4, 4), device=dev), torch.empty((b, 25), device=dev)
        jac, out12 = torch.empty((b, 12, 9), device=dev), torch.empty((b, 12), device=dev)
        k9 = intrinsic.reshape(-1)[:9].contiguous().float()
        L.check(L.lib().eg3d_pose_chain_fwd(pose.data_ptr(), translation.data_ptr(), k9.data_ptr(), b, m, float(radius), ext.data_ptr(), cam.data_ptr(),
                                            jac.data_ptr(), out12.data_ptr(), L.stream_ptr()), 'pose_chain_fwd')
        ctx.save_for_backward(jac)
        ctx.cfg = (b, m, pose.shape[1])
        return ext, cam

    @staticmethod
    def backward(ctx, d_ext, d_cam):
        from . import _lib as L
        jac, = ctx.saved_tensors
        b, m, npose = ctx.cfg
        if d_ext is None and d_cam is None:
            return None, None, None, None, None
        d_ext = d_ext.contiguous().float() if d_ext is not None else None
        d_cam = d_cam.contiguous().float() if d_cam is not None else None
        d_pose = torch.empty((b, npose), device=jac.device) if ctx.needs_input_grad[0] else None
        d_tr = torch.empty((b, 3), device=jac.device) if ctx.needs_input_grad[1] else None
        L.check(L.lib().eg3d_pose_chain_bwd(jac.data_ptr(), L.ptr(d_ext), L.ptr(d_cam), b, m, L.ptr(d_pose), L.ptr(d_tr), L.stream_ptr()), 'pose_chain_bwd')
        return d_pose, d_tr, None, None, None


POSE_CHAIN_KERNEL = os.environ.get('EG3D_POSE_CHAIN', '1') != '0'
# C3: the canonical view of the warping loss is rendered from the planes the step's main synthesis call just computed (G.synthesis's own
# cache_backbone / use_cached_backbone, triplane.py:55-63 of the reference) instead of running the backbone a second time on the same ws:
# identical planes (the backbone is a function of ws and the const noise alone), one backbone forward less per step
SHARE_BACKBONE = os.environ.get('EG3D_C3_SHARE_BACKBONE', '1') != '0'
BATCHED_WGRAD_FINISH = os.environ.get('EG3D_BATCHED_WGRAD_FINISH', '1') != '0'     # PivotalTuner: hipops.deferred_weight_grads / flush_weight_grads
GRID_SAMPLE_KERNEL = os.environ.get('EG3D_GRID_SAMPLE', '1') != '0'     # the feature warp of the warping loss on eg3d_grid_sample_nhwc_* (0: F.grid_sample)


def pose_chain(pred: torch.Tensor, translation_opt: torch.Tensor, intrinsic: torch.Tensor, radius: float, mode: str):
    """(extrinsic [B,4,4], c [B,25]) = pose_to_cam(pose_to_rotmat(pred, mode), translation_opt, intrinsic, radius): the fused launch on the
    GPU, the PyTorch composition elsewhere."""
    if POSE_CHAIN_KERNEL and pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2 and pred.shape[1] == POSE_DIMS[mode] \
            and translation_opt.shape == (pred.shape[0], 3) and intrinsic.numel() >= 9:
        return _PoseChainFn.apply(pred, translation_opt, intrinsic, radius, mode)
    return pose_to_cam(pose_to_rotmat(pred, mode), translation_opt, intrinsic, radius)


def line_plane_intersection(plane_normal, plane_point, ray_dir, ray_point, eps=1e-6):
    """training/warping_loss.py:58-72."""
    ndotu = (plane_normal * ray_dir).sum(-1, keepdim=True)
    w_vec = ray_point - plane_point
    si = -(plane_normal * w_vec).sum(-1, keepdim=True) / ndotu
    return w_vec + si * ray_dir + plane_point


_WARP_CONST = {}


def _warp_constants(init_ext):
    """(a point on the canonical image plane, world->canonical-camera matrix): functions of the constant canonical extrinsic only,
    computed once (a host->device constant and a matrix inverse would otherwise sit inside every step and break graph capture)."""
    key = (init_ext.data_ptr(), init_ext._version)
    hit = _WARP_CONST.get(key)
    if hit is None:
        with torch.no_grad():
            pt = torch.bmm(init_ext.reshape(-1, 4, 4), torch.tensor([[0., 0., 1., 1.]], device=init_ext.device).unsqueeze(-1)).squeeze(-1)[:, :3]
            hit = (pt.clone(), torch.linalg.inv(init_ext.reshape(4, 4)).clone(), init_ext)
        if len(_WARP_CONST) > 64:
            _WARP_CONST.clear()
        _WARP_CONST[key] = hit
    return hit[0], hit[1]


def _warp_consts_packed(init_ext, intrinsic):
    """The 24 constants of eg3d_warp_project_*: canonical camera centre, a point on its image plane, rows 0..2 of the world->camera
    matrix, rows 0..1 of the intrinsics -- built once on the device."""
    key = ('packed', init_ext.data_ptr(), init_ext._version, intrinsic.data_ptr(), intrinsic._version)
    hit = _WARP_CONST.get(key)
    if hit is None:
        plane_pt, w2c = _warp_constants(init_ext)
        with torch.no_grad():
            k = intrinsic.reshape(3, 3)
            packed = torch.cat([init_ext.reshape(4, 4)[:3, 3].reshape(-1), plane_pt.reshape(-1)[:3], w2c[:3, :3].reshape(-1), w2c[:3, 3].reshape(-1),
                                k[:2].reshape(-1)]).float().contiguous()
        hit = (packed, init_ext, intrinsic)
        _WARP_CONST[key] = hit
    return hit[0]


class _WarpProjectFn(torch.autograd.Function):
    """(origins [P,3], dirs [P,3], depth [P]) -> uv [P,2] in [-1,1]: the per-pixel reprojection of warping_loss.py:18-54 as one launch per
    direction (eg3d_warp_project_fwd / _bwd) instead of ~40 + ~80 ATen kernels on 16 K-element tensors."""

    @staticmethod
    def forward(ctx, o, d, depth, consts):
        from . import _lib as L
        L.require_cuda(o, d, depth, consts)
        o, d, depth = o.contiguous().float(), d.contiguous().float(), depth.contiguous().float()
        P = depth.numel()
        uv = torch.empty((P, 2), device=o.device, dtype=torch.float32)
        L.check(L.lib().eg3d_warp_project_fwd(o.data_ptr(), d.data_ptr(), depth.data_ptr(), consts.data_ptr(), uv.data_ptr(), P, L.stream_ptr()), 'warp_project_fwd')
        ctx.save_for_backward(o, d, depth, consts)
        return uv

    @staticmethod
    def backward(ctx, duv):
        from . import _lib as L
        o, d, depth, consts = ctx.saved_tensors
        duv = duv.contiguous().float()
        P = depth.numel()
        d_o, d_d, d_t = torch.empty_like(o), torch.empty_like(d), torch.empty_like(depth)
        L.check(L.lib().eg3d_warp_project_bwd(o.data_ptr(), d.data_ptr(), depth.data_ptr(), consts.data_ptr(), duv.data_ptr(), d_o.data_ptr(), d_d.data_ptr(),
                                              d_t.data_ptr(), P, L.stream_ptr()), 'warp_project_bwd')
        return d_o, d_d, d_t, None


def warp_project(o, d, depth, init_ext, intrinsic):
    """uv [P,2] of every depth pixel in the canonical view (see _WarpProjectFn)."""
    return _WarpProjectFn.apply(o.reshape(-1, 3), d.reshape(-1, 3), depth.reshape(-1), _warp_consts_packed(init_ext, intrinsic))


class _GridSampleFn(torch.autograd.Function):
    """F.grid_sample(inp, grid, mode='bilinear', padding_mode='zeros', align_corners=False) for a channels-last fp32 map
    (eg3d_grid_sample_nhwc_fwd / _bwd: one wave per output pixel over contiguous channel rows)."""

    @staticmethod
    def forward(ctx, inp, grid):
        from . import _lib as L
        N, C, Hh, Ww = inp.shape
        _, Ho, Wo, _ = grid.shape
        grid = grid.contiguous().float()
        out = hipops.empty_cl(N, C, Ho, Wo, inp.device)
        L.check(L.lib().eg3d_grid_sample_nhwc_fwd(inp.data_ptr(), grid.data_ptr(), out.data_ptr(), N, Hh, Ww, C, Ho, Wo, L.stream_ptr()), 'grid_sample_nhwc_fwd')
        ctx.save_for_backward(inp, grid)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _lib as L
        inp, grid = ctx.saved_tensors
        N, C, Hh, Ww = inp.shape
        _, Ho, Wo, _ = grid.shape
        dout = hipops.to_cl(dout.float())
        dgrid = torch.empty_like(grid)
        dinp = hipops.zeros_cl(N, C, Hh, Ww, inp.device) if ctx.needs_input_grad[0] else None
        L.check(L.lib().eg3d_grid_sample_nhwc_bwd(inp.data_ptr(), grid.data_ptr(), dout.data_ptr(), dgrid.data_ptr(), dinp.data_ptr() if dinp is not None else None,
                                                  N, Hh, Ww, C, Ho, Wo, L.stream_ptr()), 'grid_sample_nhwc_bwd')
        return dinp, dgrid


def grid_sample_bilinear(inp: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """F.grid_sample(inp, grid, mode='bilinear', align_corners=False) -- on the library's kernel for fp32 CUDA maps with C % 4 == 0."""
    if GRID_SAMPLE_KERNEL and inp.is_cuda and inp.dtype == torch.float32 and inp.shape[1] % 4 == 0 and grid.shape[0] == inp.shape[0]:
        return _GridSampleFn.apply(hipops.to_cl(inp), grid)
    return F.grid_sample(inp, grid, mode='bilinear', align_corners=False)


def warping_loss(G, ws, canonical_cam, extrinsic, init_ext, intrinsic, depth, target_feat, feat_fn, synth_kwargs=None, cl4_ok=False):
    """Depth-reprojection loss (training/warping_loss.py:6-56): render the canonical view without gradient, lift the predicted
    depth to 3-D with the predicted extrinsic, project into the canonical image, sample canonical features there and compare
    with the target's features under a foreground mask.  The target's feature map is passed in (the reference recomputes it
    every step, :35)."""
    synth_kwargs = synth_kwargs or {}
    with torch.no_grad():
        can = G.synthesis(ws.detach(), canonical_cam.detach(), noise_mode='const', force_fp32=True, **synth_kwargs)['image']
        p4 = getattr(can, '_eg3d_padded4', None)
        res = can.shape[2]
        if cl4_ok and p4 is not None and res == can.shape[3] and (res <= 256 or res % 256 == 0):
            # the 4-float-pixel image the SR head wrote, area-resized in one pass (as the step's main image): no pad / cat / layout copy
            from . import loss_nets as LN
            can = LN.image_prepare(p4, max(1, res // 256), 1.0, 0.0)
        elif res > 256:
            can = _area_resize(can, 256)
        can_feat = feat_fn(can)
    mask = (depth < depth.mean()).float()
    res = depth.shape[-1]
    o, d = G.ray_sampler(extrinsic, intrinsic.reshape(1, 3, 3), res)
    uv = warp_project(o, d, depth, init_ext, intrinsic)                            # [res*res,2]; grad -> extrinsic (through the rays), depth
    fr = target_feat.shape[-1]
    uv_f = F.interpolate(uv.reshape(1, res, res, 2).permute(0, 3, 1, 2), size=(fr, fr), mode='bilinear').permute(0, 2, 3, 1)
    warped = grid_sample_bilinear(can_feat, uv_f)
    m = F.interpolate(mask, size=(fr, fr), mode='bilinear')
    return ((warped - target_feat) * m).abs().mean()


def _area_resize(img: torch.Tensor, size: int) -> torch.Tensor:
    """F.interpolate(img, size=(size, size), mode='area') (w_projector.py:106-110,198-200).  For an integer factor that is a plain
    k x k average: avg_pool2d runs in ~5 us forward/backward where the generic adaptive-pool kernels take 65 + 42 us at 512^2."""
    h, w = img.shape[-2:]
    if h % size == 0 and w % size == 0 and h // size == w // size:
        return F.avg_pool2d(img, h // size)
    return F.interpolate(img, size=(size, size), mode='area')


# w-space projection: ws as a stride-0 view of the optimised row (fused.BroadcastRowsFn) instead of w.repeat(...).  Not in the deterministic build: a
# replayed G.synthesis (graphed.py) copies ws into a dense static buffer and sums d w per row and then over the rows, the eager call sums all layers into
# one row -- two roundings of the same exact sums, and which steps replay depends on what ran before (two runs would differ by ~4e-7 in w)
BROADCAST_WS = not hipops.L.DETERMINISTIC
REG_BRANCH = False          # the noise regulariser on a second stream (a branch of the captured graph): measured -1.5 %


class LatentProjector:
    """Phase A.  One `step()` = pose chain (optional) -> G.synthesis with grad -> [canonical no-grad forward + warping loss] ->
    feature distance + 1e5 * noise regulariser -> backward -> Adam steps -> noise renormalisation.

    `target` [N,3,H,W]: N > 1 runs N INDEPENDENT inversions as one batch (config C5: 8 images per GPU) -- per-image latent, camera,
    noise maps and Adam state, the frozen generator weights shared; per-sample modulation makes this exactly N separate trajectories
    (networks_stylegan2.py:85-88), while the 4^2 ... 64^2 layers stop being latency-bound.  With N == 1 the noise
    maps are the generator's own `noise_const` buffers, as in the reference; with N > 1 they are per-image tensors [N,1,r,r] owned by
    the projector (`noise_maps`) and fed to the layers as per-sample noise."""

    def __init__(self, G, target: torch.Tensor, *, num_steps=400, w_avg: Optional[torch.Tensor] = None, w_std: float = 1.0,
                 start_w: Optional[torch.Tensor] = None, cam: Optional[torch.Tensor] = None, optimize_pose: bool = False,
                 feature_net: Optional[Callable] = None, warp_feature_net: Optional[Callable] = None, use_warping_loss: bool = False,
                 first_inv_lr=8e-3, cam_lr=6e-7, translation_lr=2e-4, cam_preheat_steps=50, initial_noise_factor=0.05,
                 noise_ramp_length=0.75, lr_rampdown_length=0.25, lr_rampup_length=0.05, regularize_noise_weight=1e5,
                 initial_learning_rate=0.01, radius=2.7, wplus=False, synth_kwargs: Optional[dict] = None, seed: int = 0,
                 init_noise: Optional[Dict[str, torch.Tensor]] = None, use_graph: bool = False, graph_warmup: int = 2,
                 pose_net: Optional[torch.nn.Module] = None, pose_mode: str = 'quat', translation_start=None, sr_fp16: bool = False,
                 modconv_f16x1: bool = False):
        if pose_mode not in POSE_DIMS:
            raise ValueError(f'pose_mode must be one of {sorted(POSE_DIMS)}, got {pose_mode!r}')
        self.pose_mode = pose_mode
        # sr_fp16: run the super-resolution head in the reference's fp16-operand arithmetic (one product of fp16-rounded operands) also in
        # Phase A.  The reference itself passes force_fp32=True here (w_projector.py:189), so this is an OPTION for a secondary figure
        # (to be shown to keep the final-PSNR drift small), never the default.
        self.sr_fp16 = bool(sr_fp16)
        # modconv_f16x1: EVERY modulated conv on the pre-split kernels (backbone and super-resolution head) in one product of fp16-rounded
        # operands -- the arithmetic class of the reference's TF32 convolutions on its named GPU (hipops.modconv_override).  A labelled side
        # figure of bench.py, never the headline.
        self.modconv_f16x1 = bool(modconv_f16x1)
        dev = target.device
        N = self.N = int(target.shape[0])
        if N > 1 and optimize_pose:
            raise NotImplementedError('batched projection optimises latents and noise maps; the pose chain / warping loss are per image (N = 1)')
        self.use_graph, self._graph, self._graph_warmup, self.graph_capture_error = use_graph, None, graph_warmup, None
        self.G = G.eval().requires_grad_(False)
        self.dev = dev
        self.pose_net = None
        self.num_steps, self.preheat = num_steps, (cam_preheat_steps if optimize_pose else 0)
        self.w_std, self.noise_factor, self.noise_ramp = w_std, initial_noise_factor, noise_ramp_length
        self.lr_down, self.lr_up, self.lr0, self.reg_w = lr_rampdown_length, lr_rampup_length, initial_learning_rate, regularize_noise_weight
        self.radius, self.optimize_pose, self.use_warp = radius, optimize_pose, use_warping_loss
        self.synth_kwargs = dict(synth_kwargs or {})
        self.num_ws = G.backbone.num_ws
        self.feature_net = feature_net if feature_net is not None else StubFeatureNet().to(dev)
        self.warp_net = warp_feature_net if warp_feature_net is not None else self.feature_net_map
        self._warp_cl4 = os.environ.get('EG3D_C3_CL4', '1') != '0' and bool(getattr(warp_feature_net if warp_feature_net is not None else self.feature_net, 'accepts_cl4', False))    # takes [N,4,H,W] channels_last
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        # target: [1,3,H,W] in [-1,1]  ->  [0,255] at 256^2 (w_projector.py:106-110)
        self.target = target
        t255 = (target + 1) * (255 / 2)
        if t255.shape[2] > 256:
            t255 = _area_resize(t255, 256)
        self.t255 = t255                   # also the pose estimator's input, every step (w_projector.py:106-110,148)
        with torch.no_grad():
            self.target_features = self.feature_net(t255)
            self.target_warp_feat = self.warp_net(target) if use_warping_loss else None
        w_avg = torch.zeros(1, 1, G.w_dim, device=dev) if w_avg is None else w_avg.to(dev).reshape(1, 1, -1)
        start = torch.zeros_like(w_avg) if start_w is None else (start_w.to(dev) if start_w.dim() == 3 else start_w.to(dev).reshape(1, -1, G.w_dim))
        w0 = (w_avg + start)
        if wplus and w0.shape[1] == 1:
            w0 = w0.repeat(1, self.num_ws, 1)
        if w0.shape[0] == 1 and N > 1:
            w0 = w0.repeat(N, 1, 1)
        self.w_opt = w0.clone().float().requires_grad_(True)
        self.noise_bufs = {n: b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n}
        self.noise_bufs2 = {n: b for n, b in G.superresolution.named_buffers() if 'noise_const' in n}
        self.noise_maps, self._noise_inject = {}, None
        with torch.no_grad():
            for prefix, bufs in (('backbone.synthesis.', self.noise_bufs), ('superresolution.', self.noise_bufs2)):
                for nm, b in bufs.items():
                    shape = tuple(b.shape) if N == 1 else (N, 1) + tuple(b.shape)
                    src = init_noise[prefix + nm].to(dev) if init_noise is not None else torch.randn(shape, device=dev, generator=self.gen)
                    # the backbone's AND the SR head's maps are re-drawn, become leaves and sit in the latent optimiser
                    # (w_projector.py:120,126-131).  With superresolution_noise_mode='none' the SR maps get no gradient from the
                    # synthesis: theirs is the regulariser's alone (:230-237), and Adam moves them with it every step.
                    if N == 1:
                        b.copy_(src)
                        b.requires_grad = True
                    else:
                        t = src.expand(shape).contiguous().clone()
                        t.requires_grad = True
                        self.noise_maps[prefix + nm] = t
        if N == 1:
            self._all_bufs = list(self.noise_bufs.values()) + list(self.noise_bufs2.values())
            self._buf_views = None
        else:
            self._all_bufs = [t for k, t in self.noise_maps.items() if k.startswith('backbone.')] + \
                             [t for k, t in self.noise_maps.items() if not k.startswith('backbone.')]
        self._opt_bufs = list(self._all_bufs)          # every map is optimised (reg_grads below is aligned with _all_bufs)
        if N > 1:
            self._noise_inject = {k[:-len('.noise_const')]: t for k, t in self.noise_maps.items() if k.startswith('backbone.')}
            self._buf_views = [t.detach()[i, 0] for t in self._all_bufs for i in range(N)]       # [r,r] views, image-major per map
        # Adam over the latent and the noise maps: one launch of our own (regulariser gradient, update, moments for the renormalisation)
        # instead of torch's multi-tensor add + multi-tensor Adam (64 us for 3.3 MB: ~20 blocks) + a moments pass; EG3D_HIP_ADAM=0 -> torch's
        hip_adam = os.environ.get('EG3D_HIP_ADAM', '1') != '0' and torch.device(dev).type == 'cuda'
        if use_graph:       # the schedule values live on the device so that one captured step can be replayed for every step index
            # (latent-noise scale, learning rate) of a step: two views of one 2-float tensor, set before each replay by ONE copy from the
            # schedule table built below (two fill launches otherwise)
            self._sched = torch.zeros(2, device=dev)
            self._scale_t, self._lr_t = self._sched[0], self._sched[1]
            self._sched_table = None
            self._wn = torch.zeros_like(self.w_opt)
            # the renderer's stratified / importance uniforms of a step from ONE draw in front of the replay: random draws inside a captured
            # graph cost two generator-state fills per replay on top of the two draws themselves
            self._uni = None
            rk = getattr(G, 'rendering_kwargs', None) or {}
            Dc, Df = int(rk.get('depth_resolution', 0)), int(rk.get('depth_resolution_importance', 0))
            res = int(getattr(G, 'neural_rendering_resolution', 0) or 0)
            if Dc > 0 and Df > 0 and res > 0 and 'neural_rendering_resolution' not in self.synth_kwargs:
                R = res * res
                self._uni = torch.empty(N * R * (Dc + Df), device=dev)
                self._uni_views = (self._uni[:N * R * Dc].view(N, R, Dc, 1), self._uni[N * R * Dc:].view(N * R, Df))
            self.optimizer = (hipops.HipAdam if hip_adam else functools.partial(torch.optim.Adam, fused=True, capturable=True))(
                [self.w_opt] + self._opt_bufs, betas=(0.9, 0.999), lr=self._lr_t)
        else:
            self._uni = None
            self.optimizer = (hipops.HipAdam if hip_adam else functools.partial(torch.optim.Adam, fused=True))(
                [self.w_opt] + self._opt_bufs, betas=(0.9, 0.999), lr=first_inv_lr)
        self.intrinsic = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1], device=dev).unsqueeze(0)
        self.init_ext = torch.tensor([1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 2.7, 0, 0, 0, 1.], device=dev).reshape(1, 4, 4)
        self.canonical_cam = torch.cat([self.init_ext.reshape(1, 16), self.intrinsic], -1)
        self.cam = cam.to(dev) if cam is not None else self.canonical_cam.clone()
        if self.cam.shape[0] == 1 and N > 1:
            self.cam = self.cam.repeat(N, 1)
        if optimize_pose:
            # the reference predicts the pose vector (quaternion / 6-D / two angles) with a ResNet34 (scripts/resnet) fine-tuned per
            # image; without a pose_net the vector itself is the optimisable state ("ResNet34 optional stub")
            self.pose_vec = torch.tensor([POSE_INIT[pose_mode]], device=dev).requires_grad_(True)     # = init_ext rotation
            self.translation_opt = (torch.zeros(1, 3, device=dev) if translation_start is None else
                                    torch.as_tensor(translation_start, dtype=torch.float32, device=dev).reshape(1, 3).clone()).requires_grad_(True)
            # pose_net (pose_net.ResNetPose or any module image -> pose vector): the reference's per-image fine-tuned estimator,
            # cam_predictor(target_images) every step with Adam over all its parameters (w_projector.py:62,122,148-158)
            self.pose_net = pose_net
            if pose_net is not None:
                pose_net.requires_grad_(True)
            hip_small = os.environ.get('EG3D_HIP_ADAM', '1') != '0' and torch.device(dev).type == 'cuda'
            if pose_net is not None:       # 222 tensors: one multi-tensor launch (trainable conv weights are re-packed every step, no version-keyed cache involved)
                self.cam_optimizer = torch.optim.Adam(list(pose_net.parameters()), lr=cam_lr, betas=(0.9, 0.999), fused=True, capturable=use_graph)
            elif hip_small:                # one launch instead of torch's nine one-element kernels per optimiser (capturable Adam)
                self.cam_optimizer = hipops.HipAdam([self.pose_vec], lr=cam_lr, betas=(0.9, 0.999))
            else:
                self.cam_optimizer = torch.optim.Adam([self.pose_vec], lr=cam_lr, betas=(0.9, 0.999), capturable=use_graph)
            if hip_small:
                self.translation_optimizer = hipops.HipAdam([self.translation_opt], lr=translation_lr)
            else:
                self.translation_optimizer = torch.optim.Adam([self.translation_opt], lr=translation_lr, capturable=use_graph)
        self.step_idx = 0
        self.last = {}
        self._reg_stream = None
        self._one = None
        self._arena = None

    @property
    def quat(self):
        return self.pose_vec

    def feature_net_map(self, img):
        """Spatial feature map for the warping loss from the stub net's first two stages ([N,C,h,w])."""
        return self.feature_net.stages(img, upto=2)[-1]

    def _schedule(self, step):
        t = (step - self.preheat) / max(1, (self.num_steps - self.preheat))
        w_noise_scale = self.w_std * self.noise_factor * max(0.0, 1.0 - t / self.noise_ramp) ** 2
        lr_ramp = min(1.0, (1.0 - t) / self.lr_down)
        lr_ramp = 0.5 - 0.5 * np.cos(lr_ramp * np.pi)
        lr_ramp = lr_ramp * min(1.0, t / self.lr_up)
        return w_noise_scale, self.lr0 * lr_ramp

    def step(self, w_noise: Optional[torch.Tensor] = None, **step_kwargs) -> Dict[str, torch.Tensor]:
        if self.modconv_f16x1:
            with hipops.modconv_override('f16x1'):
                return self._step(w_noise, **step_kwargs)
        return self._step(w_noise, **step_kwargs)

    def _step(self, w_noise: Optional[torch.Tensor] = None, **step_kwargs) -> Dict[str, torch.Tensor]:
        """One optimisation step.  `w_noise` (unit normal, shape of w_opt) and `render_uniforms=(u1,u2)` may be injected for
        deterministic runs; otherwise they are drawn on the device.

        With `use_graph=True` the whole step (~640 kernel launches: synthesis forward and backward, loss, fused Adam, noise
        renormalisation) is captured once into a HIP graph after `graph_warmup` eager steps and replayed afterwards; the per-step
        schedule values (latent-noise scale, learning rate) and the latent noise are device tensors refreshed before each replay.
        The host then issues one graph launch per step instead of ~640 launches (~20 us of Python/ctypes each, which is about the
        GPU time of the step itself)."""
        step = self.step_idx
        w_noise_scale, lr = self._schedule(step)
        if self.use_graph:
            if step_kwargs:
                raise ValueError('use_graph: per-step synthesis kwargs cannot change between replays; pass them as synth_kwargs')
            row = (float(w_noise_scale), float(lr))
            if self._sched_table is None or (step < len(self._sched_host) and self._sched_host[step] != row):
                # the whole schedule on the device (it only depends on the step index; rebuilt if num_steps / preheat / ramps were changed since)
                self._sched_host = [tuple(float(v) for v in self._schedule(i)) for i in range(max(self.num_steps, 1) + 64)]
                self._sched_table = torch.tensor(self._sched_host, dtype=torch.float32, device=self.dev)
            if step < len(self._sched_host) and self.optimizer.param_groups[0]['lr'] is self._lr_t:
                self._sched.copy_(self._sched_table[step])
            else:
                self._scale_t.fill_(float(w_noise_scale))
                self.optimizer.param_groups[0]['lr'].fill_(float(lr))
            if w_noise is not None:
                self._wn.copy_(w_noise)
            else:
                self._wn.normal_(generator=self.gen)
            if self._uni is not None:         # every step, the eager camera-preheat ones included: they render from the same buffer
                self._uni.uniform_(generator=self.gen)
            if self._graph is not None:
                self._graph.replay()
            elif step < self.preheat:
                # camera preheat (pose chain): the latent is frozen and un-noised for these steps (w_projector.py:249-253) -- different
                # control flow, run eagerly; the steady-state step is captured afterwards
                self.last = self._step_body(self._scale_t, None, self.synth_kwargs, False)
            elif step < self.preheat + self._graph_warmup or self.graph_capture_error is not None:
                # eager: warm-up on a side stream (allocator / autograd state, lazy kernel attributes), or capture was refused
                side = torch.cuda.Stream(device=self.dev)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.last = self._step_body(self._scale_t, self._wn, self.synth_kwargs, True)
                torch.cuda.current_stream().wait_stream(side)
            else:
                torch.cuda.synchronize()
                self.optimizer.zero_grad(set_to_none=True)
                if self.optimize_pose:
                    self.cam_optimizer.zero_grad(set_to_none=True)
                    self.translation_optimizer.zero_grad(set_to_none=True)
                from . import dist as _D
                _D.assert_comm_ready()                # multi-rank: RCCL's communicator must exist before anything is captured
                graph = torch.cuda.CUDAGraph()
                try:
                    with hipops.capture_guard(), torch.cuda.graph(graph, capture_error_mode='thread_local'):   # other threads (RCCL watchdog) may touch the runtime
                        self.last = self._step_body(self._scale_t, self._wn, self.synth_kwargs, True)
                    self._graph = graph
                    graph.replay()                    # capture records without executing
                except RuntimeError as e:             # the runtime refused the capture: keep optimising eagerly (same kernels), say so
                    import warnings
                    self.graph_capture_error = e
                    warnings.warn(f'LatentProjector: HIP graph capture failed ({e}); continuing with eager launches')
                    torch.cuda.synchronize()
                    self.optimizer.zero_grad(set_to_none=True)
                    self.last = self._step_body(self._scale_t, self._wn, self.synth_kwargs, True)
            self.step_idx += 1
            return self.last
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        wn = None
        if step >= self.preheat:
            wn = w_noise.to(self.dev) if w_noise is not None else torch.randn(self.w_opt.shape, device=self.dev, generator=self.gen)
        self.last = self._step_body(w_noise_scale, wn, dict(self.synth_kwargs, **step_kwargs), step >= self.preheat)
        self.step_idx += 1
        return self.last

    def _step_body(self, w_noise_scale, wn, kw, do_step):
        if self._arena is None:
            self._arena = hipops.ZeroArena(torch.device(self.dev))
        with hipops.zero_arena(self._arena):
            return self._step_body_inner(w_noise_scale, wn, kw, do_step)

    def _step_body_inner(self, w_noise_scale, wn, kw, do_step):
        G = self.G
        if self.optimize_pose:
            pred = self.pose_net(self.t255) if self.pose_net is not None else self.pose_vec
            pred_ext, pred_cam = pose_chain(pred, self.translation_opt, self.intrinsic, self.radius, self.pose_mode)
        else:
            pred_ext, pred_cam = None, self.cam
        # The noise regulariser only reads the noise buffers.  While its kernel occupied 17 CUs for ~0.4 ms (rounds 1-2) it ran on a second
        # stream -- a parallel branch of the captured graph, joined where the loss is formed.  Since it is three multi-block passes of
        # ~50 us together the fork / join of the replayed graph costs more than it hides: in line by default (+1.5 % per step), the branch
        # stays available as the module attribute REG_BRANCH.
        cur = torch.cuda.current_stream()
        reg_branch = REG_BRANCH
        if reg_branch and self._reg_stream is None:
            self._reg_stream = torch.cuda.Stream(device=self.dev)
            _quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if _quiet is not None:      # the noise buffers' gradients are accumulated from two streams on purpose
                _quiet(False)
        if reg_branch:
            self._reg_stream.wait_stream(cur)
        with torch.cuda.stream(self._reg_stream if reg_branch else cur):
            # value AND gradient of the regulariser for all 17 buffers from one launch; the gradient is added to the buffers' .grad
            # after backward with one fused multi-tensor add (through autograd it would be 17 AccumulateGrad add launches)
            if self._buf_views is None:
                reg, reg_grads = hipops.noise_regularizer([b.detach() for b in self._all_bufs], scale=float(self.reg_w), want_grad=True)
            else:       # batched: the same kernel over the N x 17 per-image maps, gradients assembled per map tensor
                reg_grads = [torch.empty_like(t) for t in self._all_bufs]
                reg, _ = hipops.noise_regularizer(self._buf_views, scale=float(self.reg_w), want_grad=True,
                                                  grads=[g[i, 0] for g in reg_grads for i in range(self.N)])
        w = self.w_opt
        if wn is not None:          # w + wn * scale in one launch (scale: a device scalar under graph replay)
            w = torch.addcmul(w, wn, w_noise_scale) if torch.is_tensor(w_noise_scale) else torch.add(w, wn, alpha=float(w_noise_scale))
        # (one optimised row: a stride-0 view that the style bank reads in place -- no repeat copy, no row sum in the backward: fused.BroadcastRowsFn.
        #  Only of the step's own w + noise tensor: a view of the PARAMETER made inside a custom Function may not outlive the optimiser's in-place update.)
        ws = ((_fused.broadcast_rows(w, self.num_ws) if (BROADCAST_WS and w.is_cuda and w.shape[0] == 1 and w is not self.w_opt) else w.repeat(1, self.num_ws, 1))
              if w.shape[1] == 1 else w)
        if self._noise_inject is not None:
            kw = dict(kw, noise_inject=self._noise_inject)
        if self.use_graph and self._uni is not None and 'render_uniforms' not in kw:
            kw = dict(kw, render_uniforms=self._uni_views)          # drawn in front of the step (see __init__)
        share = self.use_warp and self.optimize_pose and SHARE_BACKBONE and (self.use_graph or not getattr(G, 'graph_eager', False))      # (a plain eager call keeps its own auto-captured replay: graphed.py)
        if share:           # the canonical view of the warping loss renders the SAME planes (same ws, const noise): keep them (see warping_loss)
            kw = dict(kw, cache_backbone=True)
        out = G.synthesis(ws, pred_cam, noise_mode='const', **(dict(sr_fp16=True) if self.sr_fp16 else dict(force_fp32=True)), **kw)
        from . import loss_nets as LN
        p4 = getattr(out['image'], '_eg3d_padded4', None)
        res = out['image'].shape[2]
        if p4 is not None and getattr(self.feature_net, 'accepts_cl4', False) and (res <= 256 or res % 256 == 0) and res == out['image'].shape[3]:
            img = LN.image_prepare(p4, max(1, res // 256), 127.5, 128.0)              # scale, shift, area resize and 4-float pixels in one pass
        else:
            img = out['image'] * 127.5 + 128
            if img.shape[2] > 256:
                img = _area_resize(img, 256)
        dist_i = LN.sqdist(self.feature_net(img), self.target_features)              # per image; independent trajectories: the sum's
        dist = dist_i.sum() if dist_i.numel() > 1 else dist_i.reshape(())            # gradient is each image's own gradient (one image: a view)
        if reg_branch:
            cur.wait_stream(self._reg_stream)
        loss = dist + reg                         # reported value; only `dist` (and the warping term) goes through autograd
        warp = None
        if self.use_warp and self.optimize_pose:
            kw_can = {k: v for k, v in kw.items() if k != 'cache_backbone'}
            warp = warping_loss(G, ws, self.canonical_cam, pred_ext, self.init_ext, self.intrinsic, out['image_depth'],
                                self.target_warp_feat, self.warp_net, dict(kw_can, use_cached_backbone=True) if share else kw_can,
                                cl4_ok=self._warp_cl4)
            if share:
                G._last_planes = None          # (nothing outside this step may render stale planes)
            loss = loss + warp
        self.optimizer.zero_grad(set_to_none=True)
        if self.optimize_pose:
            self.cam_optimizer.zero_grad(set_to_none=True)
            self.translation_optimizer.zero_grad(set_to_none=True)
        if self._one is None:
            self._one = torch.ones((), device=dist.device)          # the seed gradient, allocated once (backward() would fill a new one per step)
        (dist if warp is None else dist + warp).backward(gradient=self._one)
        fused_adam = isinstance(self.optimizer, hipops.HipAdam) and do_step
        if not fused_adam:
            have = [(b.grad, g) for b, g in zip(self._opt_bufs, reg_grads) if b.grad is not None]
            for b, g in zip(self._opt_bufs, reg_grads):
                if b.grad is None:                 # a backbone buffer the synthesis did not read (noise_mode overridden): regulariser only
                    b.grad = g
            if have:
                torch._foreach_add_([a for a, _ in have], [g for _, g in have])
        if self.optimize_pose:                 # order of w_projector.py:249-261
            self.cam_optimizer.step()
        if fused_adam:                         # regulariser gradient, update and renormalisation of the maps in two launches
            self.optimizer.step(extra_grads=dict(zip(self._opt_bufs, reg_grads)), normalize={b: self.N for b in self._all_bufs})
        elif do_step:
            self.optimizer.step()
        if self.optimize_pose:
            self.translation_optimizer.step()
        if not fused_adam:
            hipops.noise_normalize_(self._all_bufs if self._buf_views is None else self._buf_views)   # buf -= mean; buf *= rsqrt(mean(buf^2)) (w_projector.py:264-270)
        if not torch.cuda.is_current_stream_capturing():
            # the feature distance is accumulated into a slice of the step's ZeroArena, which the next step clears: callers that collect
            # per-step values lazily must get their own copy (under graph replay `last` is documented as static buffers)
            dist_i = dist_i.detach().clone()
            dist = dist_i.sum() if dist_i.numel() > 1 else dist_i.reshape(())
        last = dict(loss=loss.detach(), dist=dist.detach(), dist_per_image=dist_i.detach(), reg=reg.detach() if torch.is_tensor(reg) else reg, image=out['image'].detach(),
                    cam=pred_cam.detach(), ws=ws.detach())
        if warp is not None:
            last['warp'] = warp.detach()
        return last


class PivotalTuner:
    """Phase B.  One `step()` = G.synthesis(w_pivot, cam) with default kwargs (noise_mode='random') -> L2 + LPIPS at 512^2 and
    128^2 + depth TV -> backward into all generator weights -> Adam."""

    def __init__(self, G, target: torch.Tensor, w_pivot: torch.Tensor, cam: torch.Tensor, *, lr=3e-4, l2_lambda=1.0, lpips_lambda=1.0,
                 lpips_threshold=0.06, feature_net: Optional[Callable] = None, synth_kwargs: Optional[dict] = None, sr_fp16: bool = True,
                 use_graph: bool = False, graph_warmup: int = 2, device_stop: bool = True):
        """`use_graph`: after `graph_warmup` eager steps the whole step (forward, objective, backward into all weights, fused Adam: ~370
        launches) is captured into one HIP graph and replayed; steps that read the early-stop criterion (`step(early_stop=True)`, a host
        sync before the update, single_id_coach.py:68-71) run eagerly in between, so the reference's leave-before-update order is kept.
        The tensors in `last` are then static buffers that the next replay overwrites.
        `device_stop` (with the library's Adam): the criterion is ALSO evaluated on the device in every step, captured or not -- a sticky flag
        that masks the update from the step at which it is first met (`stopped()` reads it): the reference's every-step check without leaving
        the replayed graph.
        `sr_fp16` (default, as the reference: BaseCoach.forward calls G.synthesis without force_fp32, base_coach.py:162-164): the
        super-resolution head's convolutions -- forward, data and weight gradients -- run with one product of fp16-rounded operands instead
        of the three-product fp32-equivalent split; pass False (or force_fp32=True in synth_kwargs) for fp32-equivalent tuning."""
        self.G = G
        G.requires_grad_(True)
        self.target = target
        self.target_128 = _area_resize(target, G.neural_rendering_resolution)
        self.w_pivot, self.cam = w_pivot.detach(), cam.detach()
        self.l2_lambda, self.lpips_lambda, self.thr = l2_lambda, lpips_lambda, lpips_threshold
        self.device_stop = bool(device_stop)
        self.feature_net = feature_net if feature_net is not None else StubFeatureNet().to(target.device)
        with torch.no_grad():
            self.tf = self.feature_net(target)
            self.tf128 = self.feature_net(self.target_128)
            # the targets with 4-float pixels (channel 3 = 0), the layout the SR head and the renderer hand their images over in
            self.target4 = _pad_cl4(target)
            self.target4_128 = _pad_cl4(self.target_128)
        self.use_graph, self._graph, self._graph_warmup, self.graph_capture_error, self._eager_steps = bool(use_graph), None, int(graph_warmup), None, 0
        # Adam over the 30.7 M parameters.  The library's own kernel (eg3d_adam_step, hipops.HipAdam; EG3D_HIP_ADAM=0 -> torch's fused
        # multi-tensor Adam): device-side learning rate / step count AND a device-side skip flag -- the early stop of the reference's loop
        # (`if loss_lpips <= threshold: break` BEFORE the update, single_id_coach.py:68-71) then needs no host sync: the captured step
        # compares every step, sets a sticky `done` flag on the device and the update is masked from that step on (`device_stop`)
        params = [p for p in G.parameters()]
        self.hip_adam = (os.environ.get('EG3D_HIP_ADAM', '1') != '0' and target.is_cuda
                         and all(p.dtype == torch.float32 and p.is_contiguous() for p in params))
        self.done_t = torch.zeros((), device=target.device)
        if self.hip_adam:
            self.optimizer = hipops.HipAdam(params, lr=lr)
        else:
            self.optimizer = torch.optim.Adam(params, lr=lr, fused=True, capturable=self.use_graph)
            # the fused kernel updates parameters without bumping their version counters, which the packed-weight caches key on
            self.optimizer.register_step_post_hook(lambda *_: hipops.weights_changed())
        self.synth_kwargs = dict(synth_kwargs or {})
        self.synth_kwargs.setdefault('sr_fp16', bool(sr_fp16))
        self.last = {}
        self._arena = self._arena_eager = None

    def _fused_objective(self, out):
        """The same objective as the ATen composition in `_step` from five reduction launches per direction (loss_nets.weighted_objective):
        both L2 terms and both feature distances read the images with 4-float pixels straight from the SR head / the renderer, the depth
        total variation is one kernel.  None when the images did not come with that layout (CPU tensors, N > 1, a resized SR input)."""
        from . import loss_nets as LN
        img, raw, depth = out['image'], out['image_raw'], out['image_depth']
        p4, r4 = getattr(img, '_eg3d_padded4', None), getattr(raw, '_eg3d_padded4', None)
        if p4 is None or r4 is None or img.shape[0] != 1 or p4.shape != self.target4.shape or r4.shape != self.target4_128.shape:
            return None
        cl4 = getattr(self.feature_net, 'accepts_cl4', False)
        f1 = self.feature_net(p4 if cl4 else img)
        f2 = self.feature_net(r4 if cl4 else raw)
        d3 = depth.squeeze(0)                               # [1,H,W] for one image, as compute_tv_norm takes it (single_id_coach.py:81)
        if f1.shape != self.tf.shape or f2.shape != self.tf128.shape or d3.dim() != 3:
            return None
        d3 = d3.contiguous()
        B, Hh, Ww = d3.shape
        terms = [('sq', 0, 1.0 / img.numel(), p4, self.target4), ('sq', 0, 1.0 / raw.numel(), r4, self.target4_128),
                 ('sq', 1, 1.0, f1, self.tf), ('sq', 1, 1.0, f2, self.tf128), ('tv', 2, 1.0 / (B * (Hh - 1) * (Ww - 1)), d3)]
        res = LN.weighted_objective(terms, (self.l2_lambda, self.lpips_lambda, 1.0))
        if res is None:
            return None
        total, parts = res
        return total, parts.unbind(0)

    def step(self, early_stop: bool = False, **step_kwargs) -> Dict[str, torch.Tensor]:
        # every zero-initialised accumulator of the step (split-K outputs, gradient sums) comes out of one arena cleared by one launch;
        # nothing allocated from it outlives the step (gradients are consumed by optimizer.step() below)
        if self._arena is None:
            self._arena = hipops.ZeroArena(self.target.device)
        if not self.use_graph or early_stop or self.graph_capture_error is not None:
            # the captured graph holds raw pointers into self._arena.buf, which begin() re-allocates when the demand grows (an eager step
            # with other shapes or a fallback path): once a graph exists, eager steps take their accumulators from an arena of their own
            if self._graph is not None:
                if self._arena_eager is None:
                    self._arena_eager = hipops.ZeroArena(self.target.device)
                arena = self._arena_eager
            else:
                arena = self._arena
            with hipops.zero_arena(arena):
                return self._step(early_stop, **step_kwargs)
        if step_kwargs:
            raise ValueError('use_graph: per-step synthesis kwargs cannot change between replays; pass them as synth_kwargs')
        if self._graph is not None:
            self._graph.replay()
            hipops.weights_changed()              # the optimizer's post-step hook does not run under replay
            self.last = self._graph_last
            return self.last
        if self._eager_steps < self._graph_warmup:             # warm-up on a side stream (allocator / autograd state, lazy kernel attributes)
            side = torch.cuda.Stream(device=self.target.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), hipops.zero_arena(self._arena):
                res = self._step(False)
            torch.cuda.current_stream().wait_stream(side)
            self._eager_steps += 1
            return res
        torch.cuda.synchronize()
        self.optimizer.zero_grad(set_to_none=True)
        from . import dist as _D
        _D.assert_comm_ready()
        graph = torch.cuda.CUDAGraph()
        try:
            with hipops.capture_guard(), torch.cuda.graph(graph, capture_error_mode='thread_local'), hipops.zero_arena(self._arena):
                self._step(False)
            self._graph, self._graph_last = graph, self.last
            graph.replay()                        # capture records without executing
            hipops.weights_changed()
        except RuntimeError as e:                 # the runtime refused the capture: keep tuning eagerly (same kernels), say so
            import warnings
            self.graph_capture_error = e
            warnings.warn(f'PivotalTuner: HIP graph capture failed ({e}); continuing with eager launches')
            torch.cuda.synchronize()
            self.optimizer.zero_grad(set_to_none=True)
            with hipops.zero_arena(self._arena):
                return self._step(False)
        return self.last

    def _step(self, early_stop: bool = False, **step_kwargs) -> Dict[str, torch.Tensor]:
        G = self.G
        out = G.synthesis(self.w_pivot[:, :G.backbone.num_ws], self.cam[:, :25], **dict(self.synth_kwargs, **step_kwargs))
        fusedobj = self._fused_objective(out)
        if fusedobj is not None:
            loss, (l2, lp, tv) = fusedobj
        else:
            l2 = F.mse_loss(out['image'], self.target) + F.mse_loss(out['image_raw'], self.target_128)
            lp = (self.feature_net(out['image']) - self.tf).square().sum() + (self.feature_net(out['image_raw']) - self.tf128).square().sum()
            tv = compute_tv_norm(out['image_depth'].squeeze(0))
            loss = l2 * self.l2_lambda + lp * self.lpips_lambda + tv
        self.last = dict(loss=loss.detach(), l2=l2.detach(), lpips=lp.detach(), tv=tv.detach(), image=out['image'].detach(), done=False)
        self.optimizer.zero_grad(set_to_none=True)
        if early_stop and bool(lp.item() <= self.thr):            # the reference's per-step host sync; it leaves BEFORE the update
            self.last['done'] = True                              # (single_id_coach.py:68-71)
            self.done_t.fill_(1.0)
            return self.last
        if self.hip_adam and self.device_stop:
            # the same decision on the device, every step, replayable: done |= (lpips <= threshold); the update below is skipped once set
            hipops.early_stop_flag(lp.detach().reshape(()), self.thr, self.done_t)
            self.last['done_flag'] = self.done_t
        if BATCHED_WGRAD_FINISH:        # the 17 conv weight gradients leave their packed form in one launch, after the backward pass
            with hipops.deferred_weight_grads() as pending:
                loss.backward()
            hipops.flush_weight_grads(pending)
        else:
            loss.backward()
        if self.hip_adam:
            self.optimizer.step(skip=self.done_t if self.device_stop else None)
        else:
            self.optimizer.step()
        return self.last

    def stopped(self) -> bool:
        """Host read of the device-side stop flag (one synchronising copy of 4 bytes): call it every step or every k steps -- the weights are
        those of the step at which the criterion was first met either way (the update has been masked since)."""
        return bool(self.done_t.item() != 0)


def _pad_cl4(img: torch.Tensor) -> torch.Tensor:
    """[N,3,H,W] -> [N,4,H,W] channels_last, channel 3 = 0."""
    n, c, h, w = img.shape
    return torch.cat([img.float(), img.new_zeros(n, 4 - c, h, w, dtype=torch.float32)], 1).contiguous(memory_format=torch.channels_last)


def psnr_01(img: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """PSNR = -10 log10(MSE) on [0,1]-scaled images (single_id_coach.py:90-94 scaling)."""
    a = (img.clamp(-1, 1) + 1) / 2
    b = (target.clamp(-1, 1) + 1) / 2
    return -10.0 * torch.log10(F.mse_loss(a, b))
