"""
CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the inversion inner loops: a pure-PyTorch restatement of one step of

  * the latent projector      training/projectors/w_projector.py:145-270   (Phase A; optional pose chain + warping loss)
  * the pivotal-tuning coach  training/coaches/base_coach.py:101-126, single_id_coach.py:64-77   (Phase B)

on top of oracle/eg3d_oracle.py.  The perceptual networks of the reference (VGG16-LPIPS, torchvision VGG16, LPIPS-AlexNet) are
third-party weights that are not available offline, so -- exactly like the product (inv3d_amd.inversion.StubFeatureNet) -- the
loss uses a fixed-random 3-stage conv feature pyramid; everything else (schedules, loss assembly, optimiser, noise
renormalisation, pose parametrisation, line-plane reprojection) follows the cited reference lines.

Parity status: PINNED.  tests/golden/make_golden.py lifts the reference's own loop bodies out of their (un-importable: wandb /
lpips / torchvision / mrcfile) modules by AST -- the `for step in tqdm(range(num_steps))` body of w_projector.project
(w_projector.py:145-270) and the `for i in tqdm(range(max_pti_steps))` body of SingleIDCoach.train (single_id_coach.py:64-77)
together with BaseCoach.calc_loss / forward / compute_tv_norm (base_coach.py:101-126,162-164,294-305) -- executes them unmodified
around the reference's own generator classes, RaySampler and training.warping_loss.calc_warping_loss with the stub networks below
plugged in where the third-party networks go, and asserts that ProjectorOracle / PivotalTunerOracle reproduce the recorded
trajectories (losses per step, final latent / pose / translation / noise buffers / weights) for the quaternion, 6-D and Euler pose
chains (fixtures tests/golden/projector_loop.npz, tuner_loop.npz).  What remains unpinned is only the arithmetic of the third-party
perceptual networks themselves (oracle/loss_nets_oracle.py).

Used by tests/ (GPU-vs-CPU trajectory drift) and by bench.py's cpu_baseline leg.  Never imported by the product.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import eg3d_oracle as O


def stub_feature_weights(widths=(16, 32, 64), seed=1234) -> List[torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    ws, cin = [], 4
    for w in widths:
        ws.append(torch.randn(w, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
        cin = w
    return ws


def stub_features(img: torch.Tensor, ws: List[torch.Tensor]) -> torch.Tensor:
    n, c, h, w = img.shape
    x = torch.cat([img, img.new_zeros(n, 4 - c, h, w)], 1) if c < 4 else img
    feats = []
    for wt in ws:
        x = F.avg_pool2d(O.bias_act(F.conv2d(x, wt, padding=1), None, act='lrelu'), 2)
        f = x * torch.rsqrt(x.square().sum(1, keepdim=True) + 1e-10)
        feats.append(f.flatten(1) / math.sqrt(f.shape[2] * f.shape[3]))
    return torch.cat(feats, 1)


def stub_feature_map(img: torch.Tensor, ws: List[torch.Tensor]) -> torch.Tensor:
    n, c, h, w = img.shape
    x = torch.cat([img, img.new_zeros(n, 4 - c, h, w)], 1) if c < 4 else img
    for wt in ws[:2]:
        x = F.avg_pool2d(O.bias_act(F.conv2d(x, wt, padding=1), None, act='lrelu'), 2)
    return x


def pose_to_cam(rotmat, translation_opt, intrinsic, radius=2.7):
    """w_projector.py:160-172."""
    b = rotmat.shape[0]
    pred_translation = -radius * rotmat[:, :3, 2]
    t_world = -torch.bmm(rotmat, translation_opt.unsqueeze(-1)).squeeze(-1) * radius
    t = t_world + pred_translation
    t = t / torch.norm(t, dim=-1, keepdim=True) * radius
    bottom = torch.tensor([[[0., 0., 0., 1.]]]).repeat(b, 1, 1)
    ext = torch.cat([torch.cat([rotmat, t.unsqueeze(-1)], 2), bottom], 1)
    return ext, torch.cat([ext.reshape(b, 16), intrinsic.reshape(1, 9).expand(b, 9)], 1)


def line_plane_collision(plane_normal, plane_point, ray_dir, ray_point):
    """training/warping_loss.py:58-72."""
    ndotu = (plane_normal * ray_dir).sum(-1, keepdim=True)
    w_vec = ray_point - plane_point
    si = -(plane_normal * w_vec).sum(-1, keepdim=True) / ndotu
    return w_vec + si * ray_dir + plane_point


def warping_loss(P, cfg, ws, canonical_cam, extrinsic, init_ext, intrinsic, depth, target_feat, fw, u1, u2, feat_fn=None):
    """training/warping_loss.py:6-56; feat_fn(img) stands where get_features(img, torch_vgg, '14') does (default: the stub map).
    The reference runs the canonical forward and its features with autograd enabled but on detached inputs and frozen networks, so
    no gradient flows there; gradient reaches `extrinsic` and `depth` only (:20-21)."""
    feat_fn = feat_fn if feat_fn is not None else (lambda im: stub_feature_map(im, fw))
    with torch.no_grad():
        can = O.synthesis(P, cfg, ws.detach(), canonical_cam, u1, u2, noise_mode='const')['image']
        if can.shape[2] > 256:
            can = F.interpolate(can, size=(256, 256), mode='area')
        can_feat = feat_fn(can)
    mask = (depth < depth.mean()).float()
    res = depth.shape[-1]
    o, d = O.ray_sampler(extrinsic, intrinsic.reshape(1, 3, 3), res)
    xyz = (o + d * depth.reshape(1, -1, 1))[0]
    cam_o = init_ext[:, :3, 3].expand(xyz.shape[0], 3)
    plane_pt = torch.bmm(init_ext.reshape(-1, 4, 4), torch.tensor([[0., 0., 1., 1.]]).unsqueeze(-1)).squeeze(-1)[:, :3]
    hit = line_plane_collision(-cam_o, plane_pt.expand_as(cam_o), xyz - cam_o, cam_o)
    hit1 = torch.cat([hit, torch.ones(hit.shape[0], 1)], -1).t()
    uv = (torch.linalg.inv(init_ext.reshape(4, 4)) @ hit1)[:3].t()
    uv = uv / uv[:, 2:]
    uv = (intrinsic.reshape(3, 3) @ uv.t())[:2].t()
    uv = (uv - 0.5) * 2
    fr = target_feat.shape[-1]
    uv_f = F.interpolate(uv.reshape(1, res, res, 2).permute(0, 3, 1, 2), size=(fr, fr), mode='bilinear').permute(0, 2, 3, 1)
    warped = F.grid_sample(can_feat, uv_f, mode='bilinear', align_corners=False)
    m = F.interpolate(mask, size=(fr, fr), mode='bilinear')
    return ((warped - target_feat) * m).abs().mean()


class StubPoseNet(torch.nn.Module):
    """Stand-in for the ResNet-34 pose estimator (scripts/resnet/resnet.py) in the loop pins: pred = base + A . pool4x4(img / 255).
    It depends on its input, so the reference's input convention ([0,255] image area-resized to 256^2, w_projector.py:106-110,148) is
    part of what is pinned."""

    def __init__(self, base: torch.Tensor, seed: int = 7, gain: float = 0.05):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.base = torch.nn.Parameter(base.clone().float().reshape(1, -1))
        self.A = torch.nn.Parameter(torch.randn(self.base.shape[1], 3 * 16, generator=g) * gain)

    def forward(self, img):
        f = F.adaptive_avg_pool2d(img / 255.0, 4).flatten(1) - 0.5
        return self.base + f @ self.A.t()


POSE_INIT = {'quat': [0., 1., 0., 0.],                     # rotation of the canonical extrinsic diag(1,-1,-1)
             '6d': [1., 0., 0., 0., -1., 0.],
             'euler': [0., 0.]}


class ProjectorOracle:
    """CPU twin of inv3d_amd.inversion.LatentProjector (same arguments, same injected randomness); one step() = one iteration of
    the loop at w_projector.py:145-270."""

    def __init__(self, P: Dict[str, torch.Tensor], cfg: O.GenConfig, target, *, num_steps=400, cam=None, optimize_pose=False,
                 use_warping_loss=False, init_noise: Optional[Dict[str, torch.Tensor]] = None, w_start=None, wplus=False,
                 first_inv_lr=8e-3, cam_lr=6e-7, translation_lr=2e-4, cam_preheat_steps=50, initial_noise_factor=0.05,
                 noise_ramp_length=0.75, lr_rampdown_length=0.25, lr_rampup_length=0.05, regularize_noise_weight=1e5,
                 initial_learning_rate=0.01, w_std=1.0, radius=2.7, pose_mode='quat', pose_net=None, feature_fn=None,
                 warp_feature_fn=None, translation_start=None):
        self.P, self.cfg = dict(P), cfg
        self.num_steps, self.preheat = num_steps, (cam_preheat_steps if optimize_pose else 0)
        self.w_std, self.noise_factor, self.noise_ramp = w_std, initial_noise_factor, noise_ramp_length
        self.lr_down, self.lr_up, self.lr0, self.reg_w = lr_rampdown_length, lr_rampup_length, initial_learning_rate, regularize_noise_weight
        self.radius, self.optimize_pose, self.use_warp, self.pose_mode = radius, optimize_pose, use_warping_loss, pose_mode
        self.fw = stub_feature_weights()
        self.feature_fn = feature_fn if feature_fn is not None else (lambda im: stub_features(im, self.fw))
        self.warp_feature_fn = warp_feature_fn if warp_feature_fn is not None else (lambda im: stub_feature_map(im, self.fw))
        self.target = target
        t255 = (target + 1) * (255 / 2)
        if t255.shape[2] > 256:
            t255 = F.interpolate(t255, size=(256, 256), mode='area')
        self.t255 = t255                                          # w_projector.py:106-110: also what the pose estimator sees (:148)
        with torch.no_grad():
            self.target_features = self.feature_fn(t255)
            self.target_warp_feat = self.warp_feature_fn(target) if use_warping_loss else None
        w0 = torch.zeros(1, 1, cfg.w_dim) if w_start is None else w_start.reshape(1, -1, cfg.w_dim).clone()
        if wplus and w0.shape[1] == 1:
            w0 = w0.repeat(1, cfg.num_ws, 1)
        self.w_opt = w0.float().requires_grad_(True)
        # the backbone's AND the SR head's noise buffers are re-drawn, made leaves that require grad (w_projector.py:126-131) and handed to
        # the latent optimiser (:120).  The SR head runs with noise_mode 'none', so its buffers only ever receive the regulariser's
        # gradient (:230-237) -- which Adam applies every step before the renormalisation (:264-270).
        self.buf_names = [k for k in P if k.endswith('noise_const') and k.startswith('backbone.')]
        self.buf_names2 = [k for k in P if k.endswith('noise_const') and not k.startswith('backbone.')]
        for k in self.buf_names + self.buf_names2:
            v = init_noise[k].clone() if init_noise is not None else torch.randn_like(P[k])
            self.P[k] = v.requires_grad_(True)
        self.bufs = [self.P[k] for k in self.buf_names]
        self.bufs2 = [self.P[k] for k in self.buf_names2]
        self.optimizer = torch.optim.Adam([self.w_opt] + self.bufs + self.bufs2, betas=(0.9, 0.999), lr=first_inv_lr)
        self.intrinsic = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).unsqueeze(0)
        self.init_ext = torch.tensor([1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 2.7, 0, 0, 0, 1.]).reshape(1, 4, 4)
        self.canonical_cam = torch.cat([self.init_ext.reshape(1, 16), self.intrinsic], -1)
        self.cam = cam if cam is not None else self.canonical_cam.clone()
        self.pose_net = None
        if optimize_pose:
            self.pose_net = pose_net
            if pose_net is None:                                   # free pose vector ("ResNet34 optional stub")
                self.pose_vec = torch.tensor([POSE_INIT[pose_mode]]).requires_grad_(True)
                cam_params = [self.pose_vec]
            else:
                cam_params = list(pose_net.parameters())
            self.translation_opt = (torch.zeros(1, 3) if translation_start is None else
                                    torch.tensor(translation_start, dtype=torch.float32).reshape(1, 3)).requires_grad_(True)
            self.cam_optimizer = torch.optim.Adam(cam_params, lr=cam_lr, betas=(0.9, 0.999))
            self.translation_optimizer = torch.optim.Adam([self.translation_opt], lr=translation_lr)
        self.step_idx = 0
        self.last = {}

    @property
    def quat(self):
        return self.pose_vec

    def _schedule(self, step):
        t = (step - self.preheat) / max(1, (self.num_steps - self.preheat))
        w_noise_scale = self.w_std * self.noise_factor * max(0.0, 1.0 - t / self.noise_ramp) ** 2
        lr_ramp = min(1.0, (1.0 - t) / self.lr_down)
        lr_ramp = 0.5 - 0.5 * np.cos(lr_ramp * np.pi)
        lr_ramp = lr_ramp * min(1.0, t / self.lr_up)
        return w_noise_scale, self.lr0 * lr_ramp

    def step(self, u1, u2, w_noise: Optional[torch.Tensor] = None):
        step, cfg = self.step_idx, self.cfg
        w_noise_scale, lr = self._schedule(step)
        for g in self.optimizer.param_groups:
            g['lr'] = lr
        if self.optimize_pose:
            pred = self.pose_net(self.t255) if self.pose_net is not None else self.pose_vec
            rot = O.pose_to_rotmat(pred, self.pose_mode)
            pred_ext, pred_cam = pose_to_cam(rot, self.translation_opt, self.intrinsic, self.radius)
        else:
            pred_ext, pred_cam = None, self.cam
        w = self.w_opt
        if step >= self.preheat and w_noise is not None:
            w = w + w_noise * w_noise_scale
        ws = w.repeat(1, cfg.num_ws, 1) if w.shape[1] == 1 else w
        out = O.synthesis(self.P, cfg, ws, pred_cam, u1, u2, noise_mode='const')
        warp = None
        if self.use_warp and self.optimize_pose:
            warp = warping_loss(self.P, cfg, ws, self.canonical_cam, pred_ext, self.init_ext, self.intrinsic, out['image_depth'],
                                self.target_warp_feat, self.fw, u1, u2, feat_fn=self.warp_feature_fn)
        img = out['image'] * 127.5 + 128
        if img.shape[2] > 256:
            img = F.interpolate(img, size=(256, 256), mode='area')
        dist = (self.target_features - self.feature_fn(img)).square().sum()
        reg = O.noise_regularizer(self.bufs + self.bufs2)
        loss = dist + reg * self.reg_w
        if warp is not None:
            loss = loss + warp
        self.optimizer.zero_grad(set_to_none=True)
        if self.optimize_pose:
            self.cam_optimizer.zero_grad(set_to_none=True)
            self.translation_optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.optimize_pose:
            self.cam_optimizer.step()
        if step >= self.preheat:
            self.optimizer.step()
        if self.optimize_pose:
            self.translation_optimizer.step()
        with torch.no_grad():
            for b in self.bufs + self.bufs2:
                b -= b.mean()
                b *= b.square().mean().rsqrt()
        self.step_idx += 1
        self.last = dict(loss=loss.detach(), dist=dist.detach(), image=out['image'].detach(), cam=pred_cam.detach(), ws=ws.detach(),
                         reg=reg.detach(), warp=None if warp is None else warp.detach())
        return self.last


class PivotalTunerOracle:
    """CPU twin of inv3d_amd.inversion.PivotalTuner: all generator weights trainable, Adam 3e-4; one step() = one iteration of the
    loop at single_id_coach.py:64-77 (loss -> zero_grad -> LPIPS-threshold exit BEFORE the update -> backward -> Adam)."""

    def __init__(self, P, cfg, target, w_pivot, cam, *, lr=3e-4, l2_lambda=1.0, lpips_lambda=1.0, lpips_threshold=0.06, feature_fn=None):
        self.cfg = cfg
        self.P = {k: (v.clone().requires_grad_(True) if not k.endswith(O.BUFFER_SUFFIXES) else v.clone()) for k, v in P.items()}
        self.params = [v for k, v in self.P.items() if v.requires_grad]
        self.target = target
        self.target_128 = F.interpolate(target, size=(cfg.nrr, cfg.nrr), mode='area')
        self.w_pivot, self.cam = w_pivot.detach(), cam.detach()
        self.l2_lambda, self.lpips_lambda, self.thr = l2_lambda, lpips_lambda, lpips_threshold
        self.fw = stub_feature_weights()
        self.feature_fn = feature_fn if feature_fn is not None else (lambda im: stub_features(im, self.fw))
        with torch.no_grad():
            self.tf = self.feature_fn(target)
            self.tf128 = self.feature_fn(self.target_128)
        self.optimizer = torch.optim.Adam(self.params, lr=lr)
        self.last = {}

    def step(self, u1, u2, noise_mode='random', noises=None, early_stop=False):
        out = O.synthesis(self.P, self.cfg, self.w_pivot, self.cam, u1, u2, noise_mode=noise_mode, noises=noises)
        l2 = F.mse_loss(out['image'], self.target) + F.mse_loss(out['image_raw'], self.target_128)
        lp = (self.feature_fn(out['image']) - self.tf).square().sum() + \
             (self.feature_fn(out['image_raw']) - self.tf128).square().sum()
        tv = O.compute_tv_norm(out['image_depth'].squeeze(0))
        loss = l2 * self.l2_lambda + lp * self.lpips_lambda + tv
        self.last = dict(loss=loss.detach(), l2=l2.detach(), lpips=lp.detach(), tv=tv.detach(), image=out['image'].detach(), done=False)
        self.optimizer.zero_grad(set_to_none=True)
        if early_stop and float(lp) <= self.thr:
            self.last['done'] = True
            return self.last
        loss.backward()
        self.optimizer.step()
        return self.last


# ---------------------------------------------------------------------------------------------------
# Seeded inputs of the loop pins (shared by tests/golden/make_golden.py, which feeds them to the reference's lifted loops, and by
# the tests that replay the recorded trajectories through the oracle and through the HIP path)
# ---------------------------------------------------------------------------------------------------
PIN_PROJ_STEPS, PIN_PROJ_PREHEAT, PIN_TUNER_STEPS = 6, 2, 5
PIN_TRANSLATION_START = [0.02, -0.03, 0.05]       # at exactly zero the gradient along the viewing axis vanishes (Adam then follows rounding noise)
PIN_W_STD = 0.8


def pin_config(tuner: bool = False) -> O.GenConfig:
    """The reference loops hard-code 14 ws rows (w_projector.py:182,185; base_coach.py:163) and, in Phase B, the 128^2 raw image
    (base_coach.py:103): 256^2 planes with 16 channels everywhere; 16^2 -> 64^2 rendering for Phase A, 128^2 -> 512^2 for Phase B."""
    cfg = O.small_config(plane_res=256, channel_base=4096, **(dict(nrr=128, sr_in_res=128) if tuner else {}))
    assert cfg.num_ws == 14
    return cfg


def pin_target(cfg, P, seed=31):
    """[3,H,W] in [-1,1]: a render of another latent by the same generator."""
    with torch.no_grad():
        u1, u2 = O.make_uniforms(cfg, 1, seed=seed + 2)
        return O.synthesis(P, cfg, O.synth_ws(cfg, 1, seed=seed), O.synth_cameras(1, seed=seed + 1), u1, u2, noise_mode='const')['image'][0]


def pin_projector_inputs(cfg, P, mode, steps=PIN_PROJ_STEPS):
    uniforms = [O.make_uniforms(cfg, 1, seed=100 + k) for k in range(steps)]
    wns = [O._randn(f'wn{k}', 41, (1, 1, cfg.w_dim)) for k in range(steps)]
    init_noise = {k: O._randn('init.' + k, 42, v.shape) for k, v in P.items() if k.endswith('noise_const')}
    w0 = 0.3 * O._randn('w0', 43, (1, 1, cfg.w_dim))
    base = torch.tensor(POSE_INIT[mode]) + 0.05 * O._randn('pose0' + mode, 44, (len(POSE_INIT[mode]),))
    return dict(uniforms=uniforms, wns=wns, init_noise=init_noise, w0=w0, pose_base=base)


def pin_tuner_inputs(cfg, steps=None):
    PIN_TUNER_STEPS = steps if steps is not None else globals()['PIN_TUNER_STEPS']
    names = [f'backbone.synthesis.b{r}.{cv}' for r in cfg.block_resolutions for cv in (['conv1'] if r == 4 else ['conv0', 'conv1'])]
    res = lambda nm: int(nm.split('.b')[1].split('.')[0])                         # noqa: E731
    uniforms = [O.make_uniforms(cfg, 1, seed=200 + k) for k in range(PIN_TUNER_STEPS)]
    noises = [{nm: O._randn(f'tn{k}.' + nm, 45, (1, 1, res(nm), res(nm))) for nm in names} for k in range(PIN_TUNER_STEPS)]
    return dict(uniforms=uniforms, noises=noises, noise_names=names, w_pivot=O.synth_ws(cfg, 1, seed=1), cam=O.synth_cameras(1, seed=2))
