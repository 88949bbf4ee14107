"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A functional, device-agnostic, pure-PyTorch (fp32, CPU) restatement of the arithmetic on the
EG3D-inversion hot path of cvlab-kaist/3DGAN-Inversion (TriPlaneGenerator.synthesis and the ops under
it).  It is the *checker* for the HIP kernels in `3dgan-inversion_amd/csrc`; nothing in the product
path may import it (only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg do).

Parity pin: this file is validated against the reference itself, imported from /root/reference in the
build container by `tests/golden/make_golden.py`, which also writes the committed fixtures under
`tests/golden/*.npz`.  `tests/test_oracle_golden.py` re-checks the oracle against those fixtures on
any machine (the reference does not travel).

All randomness is INJECTED (stratified jitter `u1`, importance uniforms `u2`, per-layer noise),
because torch's CPU / HIP RNG streams differ.  Every function cites the reference lines it follows
(paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ---------------------------------------------------------------------------------------------------
# Config (values of SURVEY.md section 8; kept as data, never as constants in kernels)
# ---------------------------------------------------------------------------------------------------


def default_rendering_kwargs() -> dict:
    return dict(
        depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1.0,
        disparity_space_sampling=False, clamp_mode='softplus', white_back=False,
        superresolution_noise_mode='none', sr_antialias=True, c_gen_conditioning_zero=False, c_scale=1.0,
        decoder_lr_mul=1.0, avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2], density_reg=0.25,
        density_reg_p_dist=0.004, reg_type='l1')


@dataclass
class GenConfig:
    """Shape/config description of one TriPlaneGenerator (training/triplane.py:20-45)."""
    z_dim: int = 512
    c_dim: int = 25
    w_dim: int = 512
    plane_res: int = 256            # backbone img_resolution (triplane.py:40)
    plane_channels: int = 96        # 3 planes x 32 (triplane.py:40)
    channel_base: int = 32768
    channel_max: int = 512
    mapping_layers: int = 2
    nrr: int = 128                  # neural_rendering_resolution
    sr_in_res: int = 128            # SuperresolutionHybrid8XDC.input_resolution (superresolution.py:272)
    sr_channels: Tuple[int, int] = (256, 128)   # superresolution.py:274-277
    sr_clamp: Optional[float] = 256.0           # use_fp16 => conv_clamp 256 (superresolution.py:275)
    backbone_clamp: Optional[float] = None
    decoder_hidden: int = 64
    decoder_out: int = 32
    rendering: dict = field(default_factory=default_rendering_kwargs)

    @property
    def block_resolutions(self) -> List[int]:
        return [2 ** i for i in range(2, int(math.log2(self.plane_res)) + 1)]

    def channels(self, res: int) -> int:
        return min(self.channel_base // res, self.channel_max)

    @property
    def num_ws(self) -> int:
        # networks_stylegan2.py:488-500: one per conv, plus the last block's torgb
        n = 0
        for r in self.block_resolutions:
            n += 1 if r == 4 else 2
        return n + 1

    @property
    def img_resolution(self) -> int:
        return self.sr_in_res * 4


def full_config() -> GenConfig:
    return GenConfig()


def small_config(**kw) -> GenConfig:
    """Same topology, tiny widths (for tests that must finish in seconds on CPU)."""
    d = dict(plane_res=32, channel_base=256, channel_max=16, nrr=16, sr_in_res=16, sr_channels=(16, 8),
             w_dim=32, z_dim=32)
    d.update(kw)
    cfg = GenConfig(**d)
    cfg.rendering = default_rendering_kwargs()
    cfg.rendering.update(depth_resolution=12, depth_resolution_importance=12)
    return cfg


# ---------------------------------------------------------------------------------------------------
# Deterministic synthetic weights keyed by state-dict name (SURVEY.md section 8d "Synthetic inputs")
# ---------------------------------------------------------------------------------------------------


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def _randn(name, seed, shape):
    g = torch.Generator(device='cpu').manual_seed(_name_seed(name, seed))
    return torch.randn(shape, generator=g, dtype=torch.float32)


def _rand(name, seed, shape):
    g = torch.Generator(device='cpu').manual_seed(_name_seed(name, seed))
    return torch.rand(shape, generator=g, dtype=torch.float32)


def param_shapes(cfg: GenConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict schema of SURVEY.md Appendix C (names identical to the reference's)."""
    sh: Dict[str, Tuple[int, ...]] = {}

    def layer(prefix, cin, cout, k, res, noise=True, torgb=False):
        sh[f'{prefix}.weight'] = (cout, cin, k, k)
        sh[f'{prefix}.bias'] = (cout,)
        sh[f'{prefix}.affine.weight'] = (cin, cfg.w_dim)
        sh[f'{prefix}.affine.bias'] = (cin,)
        if not torgb:
            sh[f'{prefix}.noise_strength'] = ()
            sh[f'{prefix}.noise_const'] = (res, res)
            sh[f'{prefix}.resample_filter'] = (4, 4)

    def block(prefix, cin, cout, res, img_ch):
        if cin == 0:
            sh[f'{prefix}.const'] = (cout, res, res)
        else:
            layer(f'{prefix}.conv0', cin, cout, 3, res)
        layer(f'{prefix}.conv1', cout, cout, 3, res)
        layer(f'{prefix}.torgb', cout, img_ch, 1, res, torgb=True)
        sh[f'{prefix}.resample_filter'] = (4, 4)

    for r in cfg.block_resolutions:
        cin = cfg.channels(r // 2) if r > 4 else 0
        block(f'backbone.synthesis.b{r}', cin, cfg.channels(r), r, cfg.plane_channels)
    # mapping (networks_stylegan2.py:203-231)
    sh['backbone.mapping.embed.weight'] = (cfg.w_dim, cfg.c_dim)
    sh['backbone.mapping.embed.bias'] = (cfg.w_dim,)
    feats = [cfg.z_dim + cfg.w_dim] + [cfg.w_dim] * cfg.mapping_layers
    for i in range(cfg.mapping_layers):
        sh[f'backbone.mapping.fc{i}.weight'] = (feats[i + 1], feats[i])
        sh[f'backbone.mapping.fc{i}.bias'] = (feats[i + 1],)
    sh['backbone.mapping.w_avg'] = (cfg.w_dim,)
    # SR (superresolution.py:274-277)
    c0, c1 = cfg.sr_channels
    block('superresolution.block0', cfg.decoder_out, c0, cfg.sr_in_res * 2, 3)
    block('superresolution.block1', c0, c1, cfg.sr_in_res * 4, 3)
    # decoder (triplane.py:116-122)
    sh['decoder.net.0.weight'] = (cfg.decoder_hidden, cfg.decoder_out)
    sh['decoder.net.0.bias'] = (cfg.decoder_hidden,)
    sh['decoder.net.2.weight'] = (1 + cfg.decoder_out, cfg.decoder_hidden)
    sh['decoder.net.2.bias'] = (1 + cfg.decoder_out,)
    return sh


BUFFER_SUFFIXES = ('noise_const', 'resample_filter', 'w_avg')


def synth_params(cfg: GenConfig, seed: int = 0, bias_scale: float = 0.1) -> Dict[str, Tensor]:
    """Deterministic per-tensor weights.  Conv/FC weights and const ~ N(0,1); affine bias = 1;
    other biases ~ bias_scale*N(0,1) (non-zero so the bias path is exercised); noise_strength ~ U(0,0.1);
    noise_const ~ N(0,1); mapping fc weights ~ N(0,1)/0.01 (lr_multiplier 0.01, networks_stylegan2.py:110)."""
    out: Dict[str, Tensor] = {}
    f1 = torch.tensor([1., 3., 3., 1.])
    fir = torch.outer(f1, f1)
    fir = fir / fir.sum()
    for name, shape in param_shapes(cfg).items():
        if name.endswith('resample_filter'):
            t = fir.clone()
        elif name.endswith('affine.bias'):
            t = torch.ones(shape)
        elif name.endswith('noise_strength'):
            t = _rand(name, seed, shape) * 0.1
        elif name.endswith('w_avg'):
            t = torch.zeros(shape)
        elif name.endswith('.bias'):
            t = _randn(name, seed, shape) * bias_scale
        elif name.startswith('backbone.mapping.fc') and name.endswith('.weight'):
            t = _randn(name, seed, shape) / 0.01
        else:
            t = _randn(name, seed, shape)
        out[name] = t
    return out


def heavy_tailed_params(P: Dict[str, Tensor], seed: int = 0) -> Dict[str, Tensor]:
    """Trained-checkpoint STATISTICS on top of synth_params (no EG3D pickle exists in the image: VERDICT r5 item 7).  Trained StyleGAN2 weights and
    activations are heavy-tailed -- the reason modulated_conv2d pre-normalises its operands (training/networks_stylegan2.py:54-56).  In place of N(0,1):
      * every modulated conv weight [O, I, k, k] of the backbone and the super-resolution head: per-INPUT-channel log-normal gains exp(n_i), n ~ N(0,1),
        rescaled to unit mean square (a few input channels carry most of the contraction);
      * b4.const: 4 channels x 100; every affine.bias: 2 entries x 10 (styles of +-10 on two input channels per layer; x1000 / x30 drive the REFERENCE's own fp32 backward to NaN);
      * noise_strength ~ U(0, 1) instead of U(0, 0.1).
    Pure function of (names, seed): inv3d_amd.synthetic.apply_heavy_tail is the product-side twin."""
    out = dict(P)
    for name, t in P.items():
        body = name.startswith('backbone.synthesis.') or name.startswith('superresolution.')
        if body and name.endswith('.weight') and t.dim() == 4:
            g = torch.exp(_randn(name + '#gain', seed, (t.shape[1],)))
            g = g / g.square().mean().sqrt()
            out[name] = t * g[None, :, None, None]
        elif name.endswith('b4.const'):
            idx = torch.randperm(t.shape[0], generator=torch.Generator().manual_seed(_name_seed(name + '#outliers', seed)))[:4]
            t = t.clone(); t[idx] *= 100.0
            out[name] = t
        elif body and name.endswith('affine.bias'):
            idx = torch.randperm(t.shape[0], generator=torch.Generator().manual_seed(_name_seed(name + '#outliers', seed)))[:2]
            t = t.clone(); t[idx] *= 10.0
            out[name] = t
        elif name.endswith('noise_strength'):
            out[name] = _rand(name + '#heavy', seed, tuple(t.shape))
    return out


def synth_ws(cfg: GenConfig, n: int, seed: int = 1, wplus: bool = False) -> Tensor:
    if wplus:
        return 0.5 * _randn('ws+', seed, (n, cfg.num_ws, cfg.w_dim))
    return (0.5 * _randn('ws', seed, (n, 1, cfg.w_dim))).repeat(1, cfg.num_ws, 1)


def synth_cameras(n: int, seed: int = 2, radius: float = 2.7, focal: float = 4.2647) -> Tensor:
    """EG3D look-at-origin cam2world from radius `radius`; yaw U(-.35,.35), pitch U(-.25,.25) around pi/2
    (ranges of gen_videos.py:107-110); returns c[N,25] = [cam2world(16), intrinsics(9)]."""
    ang = _rand('cams', seed, (n, 2))
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


def lookat_cam2world(origin: Tensor, target: Tensor) -> Tensor:
    """cam2world with OpenCV convention, forward = normalize(target-origin), up = +y
    (same construction as utils/camera_utils.py:137-156 create_cam2world_matrix)."""
    fwd = F.normalize(target - origin, dim=0)
    up = torch.tensor([0., 1., 0.])
    right = -F.normalize(torch.linalg.cross(up, fwd), dim=0)
    up2 = F.normalize(torch.linalg.cross(fwd, right), dim=0)
    m = torch.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up2, fwd, origin
    return m


# ---------------------------------------------------------------------------------------------------
# L1 operator library restatement
# ---------------------------------------------------------------------------------------------------

# torch_utils/ops/bias_act.py:23-33  (name -> (fn, def_alpha, def_gain, ref, has_2nd_grad))
ACTIVATIONS = {
    'linear':   (lambda x, a: x,                        0.0, 1.0,          '',  False),
    'relu':     (lambda x, a: torch.relu(x),            0.0, math.sqrt(2), 'y', False),
    'lrelu':    (lambda x, a: F.leaky_relu(x, a),       0.2, math.sqrt(2), 'y', False),
    'tanh':     (lambda x, a: torch.tanh(x),            0.0, 1.0,          'y', True),
    'sigmoid':  (lambda x, a: torch.sigmoid(x),         0.0, 1.0,          'y', True),
    'elu':      (lambda x, a: F.elu(x),                 0.0, 1.0,          'y', True),
    'selu':     (lambda x, a: F.selu(x),                0.0, 1.0,          'y', True),
    'softplus': (lambda x, a: F.softplus(x),            0.0, 1.0,          'y', True),
    'swish':    (lambda x, a: torch.sigmoid(x) * x,     0.0, math.sqrt(2), 'x', True),
}


def bias_act(x: Tensor, b: Optional[Tensor] = None, dim: int = 1, act: str = 'linear', alpha=None, gain=None,
             clamp=None) -> Tensor:
    """y = clamp(act(x + b) * gain).  torch_utils/ops/bias_act.py:93-122 (_bias_act_ref)."""
    fn, def_alpha, def_gain, _, _ = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = fn(x, alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def setup_filter(f=(1, 3, 3, 1), normalize=True, flip_filter=False, gain=1.0, separable=None) -> Tensor:
    """torch_utils/ops/upfirdn2d.py:72-116."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = list(padding)
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return padding


def _xy(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def upfirdn2d(x: Tensor, f: Optional[Tensor], up=1, down=1, padding=0, flip_filter=False, gain=1.0) -> Tensor:
    """Zero-insert upsample -> pad/crop -> FIR -> decimate.  torch_utils/ops/upfirdn2d.py:169-213."""
    n, c, h, w = x.shape
    upx, upy = _xy(up)
    dnx, dny = _xy(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    z = x.new_zeros(n, c, h * upy, w * upx)
    z[:, :, ::upy, ::upx] = x
    z = F.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        z = F.conv2d(z, k[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        z = F.conv2d(z, k[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        z = F.conv2d(z, k[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return z[:, :, ::dny, ::dnx]


def filtered_lrelu(x: Tensor, fu: Optional[Tensor] = None, fd: Optional[Tensor] = None, b: Optional[Tensor] = None, up: int = 1, down: int = 1,
                   padding=0, gain: float = math.sqrt(2), slope: float = 0.2, clamp=None, flip_filter: bool = False) -> Tensor:
    """bias -> upsample FIR (gain up^2, all of the padding) -> leaky ReLU * gain, clamp -> downsample FIR.
    torch_utils/ops/filtered_lrelu.py:123-155 (_filtered_lrelu_ref)."""
    x = bias_act(x, b)
    x = upfirdn2d(x, fu, up=up, padding=_pad4(padding), gain=float(up) ** 2, flip_filter=flip_filter)
    x = bias_act(x, None, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    return upfirdn2d(x, fd, down=down, flip_filter=flip_filter)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1.0):
    """torch_utils/ops/upfirdn2d.py:315-350."""
    upx, upy = _xy(up)
    px0, px1, py0, py1 = _pad4(padding)
    fh, fw = (f.shape[0], f.shape[-1]) if f is not None else (1, 1)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1.0):
    """torch_utils/ops/upfirdn2d.py:354-389."""
    dx, dy = _xy(down)
    px0, px1, py0, py1 = _pad4(padding)
    fh, fw = (f.shape[0], f.shape[-1]) if f is not None else (1, 1)
    p = [px0 + (fw - dx + 1) // 2, px1 + (fw - dx) // 2, py0 + (fh - dy + 1) // 2, py1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def filter2d(x, f, padding=0, flip_filter=False, gain=1.0):
    """torch_utils/ops/upfirdn2d.py:279-311."""
    px0, px1, py0, py1 = _pad4(padding)
    fh, fw = (f.shape[0], f.shape[-1]) if f is not None else (1, 1)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """torch_utils/ops/conv2d_resample.py:31-43 (conv2d == correlation; flip_weight=False => true convolution)."""
    if not flip_weight and (w.shape[-1] > 1 or w.shape[-2] > 1):
        w = w.flip([2, 3])
    if transpose:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return F.conv2d(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """Conv with fused up/down-sampling; branch structure of torch_utils/ops/conv2d_resample.py:48-143."""
    cout, cin_g, kh, kw = w.shape
    fh, fw = (f.shape[0], f.shape[-1]) if f is not None else (1, 1)
    px0, px1, py0, py1 = _pad4(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2; py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2; py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2
    if kw == 1 and kh == 1 and down > 1 and up == 1:                               # :96
        x = upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)
    if kw == 1 and kh == 1 and up > 1 and down == 1:                               # :102
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                                                       # :108
        x = upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                                                     # :114-131
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * cin_g, cout // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:   # :134-136
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    x = upfirdn2d(x, (f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)                         # :139-143
    if down > 1:
        x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x


def fma(a, b, c):
    """torch_utils/ops/fma.py:17."""
    return torch.addcmul(c, a, b)


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    """training/networks_stylegan2.py:34-91 (fp32 branch; the fp16 pre-normalisation :54-56 is not taken in fp32)."""
    n = x.shape[0]
    cout, cin, kh, kw = weight.shape
    w = dcoefs = None
    if demodulate or fused_modconv:
        w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
    if demodulate:
        dcoefs = (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()
    if demodulate and fused_modconv:
        w = w * dcoefs.reshape(n, -1, 1, 1, 1)
    if not fused_modconv:
        x = x * styles.reshape(n, -1, 1, 1)
        x = conv2d_resample(x, weight, f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma(x, dcoefs.reshape(n, -1, 1, 1), noise)
        elif demodulate:
            x = x * dcoefs.reshape(n, -1, 1, 1)
        elif noise is not None:
            x = x + noise
        return x
    x = x.reshape(1, -1, *x.shape[2:])
    w = w.reshape(-1, cin, kh, kw)
    x = conv2d_resample(x, w, f=resample_filter, up=up, down=down, padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x + noise
    return x


def fully_connected(x, weight, bias, lr_multiplier=1.0, activation='linear'):
    """training/networks_stylegan2.py:114-127."""
    w = weight * (lr_multiplier / math.sqrt(weight.shape[1]))
    b = bias
    if b is not None and lr_multiplier != 1:
        b = b * lr_multiplier
    if activation == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return bias_act(x.matmul(w.t()), b, act=activation)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """training/networks_stylegan2.py:27-28."""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


# ---------------------------------------------------------------------------------------------------
# L2 generator graph restatement
# ---------------------------------------------------------------------------------------------------


def mapping(P, cfg: GenConfig, z, c, truncation_psi=1.0, truncation_cutoff=None):
    """MappingNetwork.forward, training/networks_stylegan2.py:232-268, through TriPlaneGenerator.mapping
    (triplane.py:48-51: c is scaled by c_scale / zeroed by c_gen_conditioning_zero)."""
    if cfg.rendering.get('c_gen_conditioning_zero', False):
        c = torch.zeros_like(c)
    c = c * cfg.rendering.get('c_scale', 0)
    pre = 'backbone.mapping'
    x = normalize_2nd_moment(z.float())
    y = normalize_2nd_moment(fully_connected(c.float(), P[f'{pre}.embed.weight'], P[f'{pre}.embed.bias']))
    x = torch.cat([x, y], 1)
    for i in range(cfg.mapping_layers):
        x = fully_connected(x, P[f'{pre}.fc{i}.weight'], P[f'{pre}.fc{i}.bias'], lr_multiplier=0.01, activation='lrelu')
    x = x.unsqueeze(1).repeat(1, cfg.num_ws, 1)
    if truncation_psi != 1:
        w_avg = P[f'{pre}.w_avg']
        if truncation_cutoff is None:
            x = w_avg.lerp(x, truncation_psi)
        else:
            x = x.clone()
            x[:, :truncation_cutoff] = w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
    return x


def synthesis_layer(P, pre, x, w, up, noise_mode, noise=None, conv_clamp=None, fused_modconv=True, gain=1.0):
    """SynthesisLayer.forward, training/networks_stylegan2.py:311-330.  `noise` (injected) replaces the
    randn of :318-319 when noise_mode == 'random' (shape [N,1,res,res], unit normal)."""
    styles = fully_connected(w, P[f'{pre}.affine.weight'], P[f'{pre}.affine.bias'])
    nz = None
    if noise_mode == 'random':
        nz = noise * P[f'{pre}.noise_strength']
    elif noise_mode == 'const':
        nz = P[f'{pre}.noise_const'] * P[f'{pre}.noise_strength']
    x = modulated_conv2d(x, P[f'{pre}.weight'], styles, noise=nz, up=up, padding=1,
                         resample_filter=P[f'{pre}.resample_filter'], flip_weight=(up == 1), fused_modconv=fused_modconv)
    act_clamp = conv_clamp * gain if conv_clamp is not None else None
    return bias_act(x, P[f'{pre}.bias'], act='lrelu', gain=math.sqrt(2) * gain, clamp=act_clamp)


def torgb_layer(P, pre, x, w, conv_clamp=None, fused_modconv=True):
    """ToRGBLayer.forward, training/networks_stylegan2.py:353-357."""
    cin = P[f'{pre}.weight'].shape[1]
    styles = fully_connected(w, P[f'{pre}.affine.weight'], P[f'{pre}.affine.bias']) * (1 / math.sqrt(cin))
    x = modulated_conv2d(x, P[f'{pre}.weight'], styles, demodulate=False, fused_modconv=fused_modconv)
    return bias_act(x, P[f'{pre}.bias'], clamp=conv_clamp)


def synthesis_block(P, pre, x, img, ws, has_conv0, noise_mode, noises=None, conv_clamp=None, fused_modconv=True, up0=2):
    """SynthesisBlock.forward ('skip' architecture), training/networks_stylegan2.py:417-461.
    ws: [N, num_conv+1, w_dim].  noises: dict layer-prefix -> [N,1,res,res] for noise_mode='random'.
    up0=1: SynthesisBlockNoUp.forward (training/superresolution.py:210-253): conv0 without up-sampling, the skip image added as it is."""
    nz = noises or {}
    i = 0
    if not has_conv0:
        x = P[f'{pre}.const'].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
        x = synthesis_layer(P, f'{pre}.conv1', x, ws[:, i], 1, noise_mode, nz.get(f'{pre}.conv1'), conv_clamp, fused_modconv); i += 1
    else:
        x = synthesis_layer(P, f'{pre}.conv0', x, ws[:, i], up0, noise_mode, nz.get(f'{pre}.conv0'), conv_clamp, fused_modconv); i += 1
        x = synthesis_layer(P, f'{pre}.conv1', x, ws[:, i], 1, noise_mode, nz.get(f'{pre}.conv1'), conv_clamp, fused_modconv); i += 1
    if img is not None and up0 == 2:
        img = upsample2d(img, P[f'{pre}.resample_filter'])
    y = torgb_layer(P, f'{pre}.torgb', x, ws[:, i], conv_clamp, fused_modconv)
    img = img + y if img is not None else y
    return x, img


def backbone_synthesis(P, cfg: GenConfig, ws, noise_mode='const', noises=None, fused_modconv=True):
    """SynthesisNetwork.forward, training/networks_stylegan2.py:503-518 -> planes [N,96,R,R]."""
    x = img = None
    w_idx = 0
    for r in cfg.block_resolutions:
        nconv = 1 if r == 4 else 2
        cur = ws[:, w_idx: w_idx + nconv + 1]
        w_idx += nconv
        x, img = synthesis_block(P, f'backbone.synthesis.b{r}', x, img, cur, r != 4, noise_mode, noises,
                                 cfg.backbone_clamp, fused_modconv)
    return img


def superresolution(P, cfg: GenConfig, rgb, x, ws, noise_mode='none', noises=None, fused_modconv=True):
    """SuperresolutionHybrid8XDC.forward, training/superresolution.py:279-290."""
    ws3 = ws[:, -1:, :].repeat(1, 3, 1)
    if x.shape[-1] != cfg.sr_in_res:
        aa = cfg.rendering.get('sr_antialias', True)
        x = F.interpolate(x, size=(cfg.sr_in_res, cfg.sr_in_res), mode='bilinear', align_corners=False, antialias=aa)
        rgb = F.interpolate(rgb, size=(cfg.sr_in_res, cfg.sr_in_res), mode='bilinear', align_corners=False, antialias=aa)
    x, rgb = synthesis_block(P, 'superresolution.block0', x, rgb, ws3, True, noise_mode, noises, cfg.sr_clamp, fused_modconv)
    x, rgb = synthesis_block(P, 'superresolution.block1', x, rgb, ws3, True, noise_mode, noises, cfg.sr_clamp, fused_modconv)
    return rgb


# The reference's five super-resolution heads (training/superresolution.py:29-58, 62-90, 94-122, 126-152, 262-290), stand-alone.
#   kind: (input resolution, (block0, block1) widths, up factor of block0.conv0, output resolution, when the inputs are resized, antialias follows sr_antialias)
SR_HEADS = {'8XDC': (128, (256, 128), 2, 512, 'ne', True), '8X': (128, (128, 64), 2, 512, 'ne', True), '4X': (128, (128, 64), 1, 256, 'lt', True),
            '2X': (64, (128, 64), 1, 128, 'ne', True), 'Deepfp32': (128, (128, 64), 1, 256, 'lt', False)}


def sr_head_param_shapes(kind: str, w_dim: int = 512, channels: int = 32) -> Dict[str, Tuple[int, ...]]:
    in_res, (c0, c1), up0, out_res, _, _ = SR_HEADS[kind]
    sh: Dict[str, Tuple[int, ...]] = {}
    for pre, cin, cout, res in (('superresolution.block0', channels, c0, in_res * up0), ('superresolution.block1', c0, c1, out_res)):
        for ly, ci, co, k in (('conv0', cin, cout, 3), ('conv1', cout, cout, 3), ('torgb', cout, 3, 1)):
            sh[f'{pre}.{ly}.weight'] = (co, ci, k, k)
            sh[f'{pre}.{ly}.bias'] = (co,)
            sh[f'{pre}.{ly}.affine.weight'] = (ci, w_dim)
            sh[f'{pre}.{ly}.affine.bias'] = (ci,)
            if ly != 'torgb':
                sh[f'{pre}.{ly}.noise_strength'] = ()
                sh[f'{pre}.{ly}.noise_const'] = (res, res)
                sh[f'{pre}.{ly}.resample_filter'] = (4, 4)
        sh[f'{pre}.resample_filter'] = (4, 4)
    if kind != '8XDC':
        sh['superresolution.resample_filter'] = (4, 4)          # registered by the 8X / 4X / 2X / Deepfp32 heads themselves (:43, :75, :108, :140)
    return sh


def sr_head_params(kind: str, seed: int = 0, w_dim: int = 512, channels: int = 32, weight_scale: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic weights of one head, same conventions as synth_params."""
    f1 = torch.tensor([1., 3., 3., 1.])
    fir = torch.outer(f1, f1)
    fir = fir / fir.sum()
    out = {}
    for name, shape in sr_head_param_shapes(kind, w_dim, channels).items():
        key = f'sr{kind}.{name}'
        if name.endswith('resample_filter'):
            t = fir.clone()
        elif name.endswith('affine.bias'):
            t = torch.ones(shape)
        elif name.endswith('noise_strength'):
            t = _rand(key, seed, shape) * 0.1
        elif name.endswith('.bias'):
            t = _randn(key, seed, shape) * 0.1
        else:
            t = _randn(key, seed, shape) * (weight_scale if name.endswith('.weight') and 'affine' not in name else 1.0)
        out[name] = t
    return out


def sr_head(P, kind: str, rgb, x, ws, sr_antialias: bool = True, conv_clamp: Optional[float] = 256.0, noise_mode='none', noises=None, fused_modconv=True):
    """forward() of the head `kind` on (rgb [N,3,r,r], x [N,32,r,r], ws [N,L,w_dim])."""
    in_res, _, up0, _, rule, follows = SR_HEADS[kind]
    ws3 = ws[:, -1:, :].repeat(1, 3, 1)
    if (x.shape[-1] != in_res) if rule == 'ne' else (x.shape[-1] < in_res):
        aa = bool(sr_antialias) if follows else False
        x = F.interpolate(x, size=(in_res, in_res), mode='bilinear', align_corners=False, antialias=aa)
        rgb = F.interpolate(rgb, size=(in_res, in_res), mode='bilinear', align_corners=False, antialias=aa)
    x, rgb = synthesis_block(P, 'superresolution.block0', x, rgb, ws3, True, noise_mode, noises, conv_clamp, fused_modconv, up0=up0)
    x, rgb = synthesis_block(P, 'superresolution.block1', x, rgb, ws3, True, noise_mode, noises, conv_clamp, fused_modconv)
    return rgb


# ---- volumetric rendering ---------------------------------------------------------------------------


def ray_sampler(cam2world: Tensor, intrinsics: Tensor, resolution: int):
    """RaySampler.forward, training/volumetric_rendering/ray_sampler.py:24-73 (need_cam_space=False)."""
    n = cam2world.shape[0]
    cam_locs = cam2world[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0], intrinsics[:, 1, 1]
    cx, cy, sk = intrinsics[:, 0, 2], intrinsics[:, 1, 2], intrinsics[:, 0, 1]
    ar = torch.arange(resolution, dtype=torch.float32, device=cam2world.device) * (1. / resolution) + (0.5 / resolution)
    y_cam = ar.repeat_interleave(resolution)[None].repeat(n, 1)     # row index  (slow)
    x_cam = ar.repeat(resolution)[None].repeat(n, 1)                # col index  (fast), :46-48
    z_cam = torch.ones_like(x_cam)
    u = lambda t: t.unsqueeze(-1)
    x_lift = (x_cam - u(cx) + u(cy) * u(sk) / u(fy) - u(sk) * y_cam / u(fy)) / u(fx) * z_cam
    y_lift = (y_cam - u(cy)) / u(fy) * z_cam
    pts = torch.stack((x_lift, y_lift, z_cam, torch.ones_like(z_cam)), dim=-1)
    world = torch.bmm(cam2world, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    dirs = F.normalize(world - cam_locs[:, None, :], dim=2)
    origins = cam_locs.unsqueeze(1).repeat(1, dirs.shape[1], 1)
    return origins, dirs


def calculate_xyz_of_depth(ray_origin, ray_dirs, depth):
    """RaySampler.calculate_xyz_of_depth, ray_sampler.py:75-93 (batch 1)."""
    res = depth.shape[-1]
    o = ray_origin.squeeze(0).reshape(res, res, 3).permute(2, 0, 1)
    d = ray_dirs.squeeze(0).reshape(res, res, 3).permute(2, 0, 1)
    xyz = o + d * depth.squeeze(0)
    return torch.cat([xyz, torch.ones(1, res, res, device=xyz.device)], 0).reshape(4, res * res)


PLANE_AXES = torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                           [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                           [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)   # renderer.py:23-37


def sample_from_planes(planes: Tensor, coords: Tensor, box_warp: float) -> Tensor:
    """renderer.py:39-66.  planes [N,3,C,H,W], coords [N,M,3] -> [N,3,M,C]."""
    n, npl, c, h, w = planes.shape
    m = coords.shape[1]
    coords = (2 / box_warp) * coords
    inv = torch.linalg.inv(PLANE_AXES.to(coords.device))
    proj = torch.bmm(coords.unsqueeze(1).expand(-1, npl, -1, -1).reshape(n * npl, m, 3),
                     inv.unsqueeze(0).expand(n, -1, -1, -1).reshape(n * npl, 3, 3))[..., :2]
    out = F.grid_sample(planes.reshape(n * npl, c, h, w), proj.unsqueeze(1).float(), mode='bilinear',
                        padding_mode='zeros', align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(n, npl, m, c)


def osg_decoder(P, feats: Tensor, lr_mul: float = 1.0):
    """OSGDecoder.forward, training/triplane.py:124-136.  feats [N,3,M,C] -> rgb [N,M,32], sigma [N,M,1]."""
    x = feats.mean(1)
    n, m, c = x.shape
    x = x.reshape(n * m, c)
    x = fully_connected(x, P['decoder.net.0.weight'], P['decoder.net.0.bias'], lr_multiplier=lr_mul)
    x = F.softplus(x)
    x = fully_connected(x, P['decoder.net.2.weight'], P['decoder.net.2.bias'], lr_multiplier=lr_mul)
    x = x.reshape(n, m, -1)
    rgb = torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001
    return rgb, x[..., 0:1]


def run_model(P, planes, coords, opts):
    """ImportanceRenderer.run_model, renderer.py:197-203 (density_noise = 0)."""
    feats = sample_from_planes(planes, coords, opts['box_warp'])
    return osg_decoder(P, feats, opts.get('decoder_lr_mul', 1.0))


def ray_march(colors, densities, depths, opts):
    """MipRayMarcher2.run_forward, training/volumetric_rendering/ray_marcher.py:25-57."""
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    colors_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
    dens_mid = (densities[:, :, :-1] + densities[:, :, 1:]) / 2
    depths_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
    assert opts['clamp_mode'] == 'softplus'
    dens_mid = F.softplus(dens_mid - 1)
    alpha = 1 - torch.exp(-(dens_mid * deltas))
    shifted = torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2)
    weights = alpha * torch.cumprod(shifted, -2)[:, :, :-1]
    rgb = torch.sum(weights * colors_mid, -2)
    wtot = weights.sum(2)
    depth = torch.sum(weights * depths_mid, -2) / wtot
    depth = torch.nan_to_num(depth, float('inf'))
    depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
    if opts.get('white_back', False):
        rgb = rgb + 1 - wtot
    return rgb * 2 - 1, depth, weights


def sample_stratified(n, m, ray_start, ray_end, depth_resolution, disparity, u1: Tensor):
    """ImportanceRenderer.sample_stratified, renderer.py:224-247.  u1 [N,M,D,1] replaces rand_like."""
    d = depth_resolution
    if disparity:
        t = torch.linspace(0, 1, d).reshape(1, 1, d, 1).repeat(n, m, 1, 1)
        t = t + u1 * (1 / (d - 1))
        return 1. / (1. / ray_start * (1. - t) + 1. / ray_end * t)
    if isinstance(ray_start, Tensor):
        steps = torch.arange(d, dtype=torch.float32) / (d - 1)                  # math_utils.py:101-118
        t = (ray_start[None] + steps.reshape(-1, 1, 1, 1) * (ray_end - ray_start)[None]).permute(1, 2, 0, 3)
        return t + u1 * ((ray_end - ray_start) / (d - 1))[..., None]
    t = torch.linspace(ray_start, ray_end, d).reshape(1, 1, d, 1).repeat(n, m, 1, 1)
    return t + u1 * ((ray_end - ray_start) / (d - 1))


def sample_pdf(bins, weights, n_importance, u: Tensor, eps=1e-5, debug: Optional[dict] = None):
    """ImportanceRenderer.sample_pdf, renderer.py:269-308 (u injected; det=True is u=linspace(0,1,N)).  debug: receives cdf, inds, below, above."""
    n_rays, ns = weights.shape
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, ns)
    if debug is not None:
        debug.update(cdf=cdf, inds=inds, below=below, above=above)
    idx = torch.stack([below, above], -1).view(n_rays, 2 * n_importance)
    cdf_g = torch.gather(cdf, 1, idx).view(n_rays, n_importance, 2)
    bins_g = torch.gather(bins, 1, idx).view(n_rays, n_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def sample_importance(z_vals, weights, n_importance, u2: Tensor, debug: Optional[dict] = None):
    """ImportanceRenderer.sample_importance, renderer.py:249-267 (runs under no_grad there)."""
    with torch.no_grad():
        b, r, s, _ = z_vals.shape
        z = z_vals.reshape(b * r, s)
        w = weights.reshape(b * r, -1)
        w = F.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
        w = F.avg_pool1d(w, 2, 1).squeeze(1)
        w = w + 0.01
        z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
        return sample_pdf(z_mid, w[:, 1:-1], n_importance, u2, debug=debug).detach().reshape(b, r, n_importance, 1)


def unify_samples(d1, c1, s1, d2, c2, s2):
    """ImportanceRenderer.unify_samples, renderer.py:212-222.  (stable sort: coarse precedes fine at ties.)"""
    d = torch.cat([d1, d2], -2)
    c = torch.cat([c1, c2], -2)
    s = torch.cat([s1, s2], -2)
    _, idx = torch.sort(d, dim=-2, stable=True)
    return (torch.gather(d, -2, idx), torch.gather(c, -2, idx.expand(-1, -1, -1, c.shape[-1])),
            torch.gather(s, -2, idx))


def get_ray_limits_box(rays_o, rays_d, box_side_length):
    """math_utils.py:46-98."""
    shp = rays_o.shape
    o = rays_o.detach().reshape(-1, 3)
    d = rays_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    bounds = torch.tensor([[-half] * 3, [half] * 3], dtype=o.dtype)
    valid = torch.ones(o.shape[0], dtype=torch.bool)
    inv = 1 / d
    sign = (inv < 0).long()
    tmin = (bounds.index_select(0, sign[:, 0])[:, 0] - o[:, 0]) * inv[:, 0]
    tmax = (bounds.index_select(0, 1 - sign[:, 0])[:, 0] - o[:, 0]) * inv[:, 0]
    tymin = (bounds.index_select(0, sign[:, 1])[:, 1] - o[:, 1]) * inv[:, 1]
    tymax = (bounds.index_select(0, 1 - sign[:, 1])[:, 1] - o[:, 1]) * inv[:, 1]
    valid[torch.logical_or(tmin > tymax, tymin > tmax)] = False
    tmin = torch.max(tmin, tymin)
    tmax = torch.min(tmax, tymax)
    tzmin = (bounds.index_select(0, sign[:, 2])[:, 2] - o[:, 2]) * inv[:, 2]
    tzmax = (bounds.index_select(0, 1 - sign[:, 2])[:, 2] - o[:, 2]) * inv[:, 2]
    valid[torch.logical_or(tmin > tzmax, tzmin > tmax)] = False
    tmin = torch.max(tmin, tzmin)
    tmax = torch.min(tmax, tzmax)
    tmin[~valid] = -1
    tmax[~valid] = -2
    return tmin.reshape(*shp[:-1], 1), tmax.reshape(*shp[:-1], 1)


def render(P, planes, origins, dirs, opts, u1: Tensor, u2: Optional[Tensor]):
    """ImportanceRenderer.forward, renderer.py:143-195.  planes [N,3,C,H,W]; u1 [N,M,D,1]; u2 [N*M,Dimp].
    Returns rgb [N,M,32], depth [N,M,1], weights_sum [N,M,1]."""
    n, m, _ = origins.shape
    if opts['ray_start'] == opts['ray_end'] == 'auto':
        rs, re = get_ray_limits_box(origins, dirs, opts['box_warp'])
        ok = re > rs
        if bool(torch.any(ok)):
            rs = rs.clone(); re = re.clone()
            rs[~ok] = rs[ok].min()
            re[~ok] = rs[ok].max()
        depths_c = sample_stratified(n, m, rs, re, opts['depth_resolution'], opts['disparity_space_sampling'], u1)
    else:
        depths_c = sample_stratified(n, m, opts['ray_start'], opts['ray_end'], opts['depth_resolution'],
                                     opts['disparity_space_sampling'], u1)
    s = depths_c.shape[2]
    coords = (origins.unsqueeze(-2) + depths_c * dirs.unsqueeze(-2)).reshape(n, -1, 3)
    col_c, den_c = run_model(P, planes, coords, opts)
    col_c = col_c.reshape(n, m, s, -1)
    den_c = den_c.reshape(n, m, s, 1)
    n_imp = opts['depth_resolution_importance']
    if n_imp > 0:
        _, _, w = ray_march(col_c, den_c, depths_c, opts)
        depths_f = sample_importance(depths_c, w, n_imp, u2)
        coords = (origins.unsqueeze(-2) + depths_f * dirs.unsqueeze(-2)).reshape(n, -1, 3)
        col_f, den_f = run_model(P, planes, coords, opts)
        col_f = col_f.reshape(n, m, n_imp, -1)
        den_f = den_f.reshape(n, m, n_imp, 1)
        d_all, c_all, s_all = unify_samples(depths_c, col_c, den_c, depths_f, col_f, den_f)
        rgb, depth, w = ray_march(c_all, s_all, d_all, opts)
    else:
        rgb, depth, w = ray_march(col_c, den_c, depths_c, opts)
    return rgb, depth, w.sum(2)


def synthesis(P, cfg: GenConfig, ws, c, u1, u2, noise_mode='const', noises=None, sr_noises=None,
              fused_modconv=True, nrr: Optional[int] = None, planes: Optional[Tensor] = None):
    """TriPlaneGenerator.synthesis, training/triplane.py:53-90 -> dict(image, image_raw, image_depth, planes)."""
    nrr = cfg.nrr if nrr is None else nrr
    cam2world = c[:, :16].view(-1, 4, 4)
    intr = c[:, 16:25].view(-1, 3, 3)
    origins, dirs = ray_sampler(cam2world, intr, nrr)
    n = origins.shape[0]
    if planes is None:
        planes = backbone_synthesis(P, cfg, ws, noise_mode, noises, fused_modconv)
    pl = planes.view(n, 3, cfg.plane_channels // 3, planes.shape[-2], planes.shape[-1])
    feat, depth, _ = render(P, pl, origins, dirs, cfg.rendering, u1, u2)
    feat_img = feat.permute(0, 2, 1).reshape(n, feat.shape[-1], nrr, nrr).contiguous()
    depth_img = depth.permute(0, 2, 1).reshape(n, 1, nrr, nrr)
    rgb = feat_img[:, :3]
    sr = superresolution(P, cfg, rgb, feat_img, ws, cfg.rendering['superresolution_noise_mode'], sr_noises, fused_modconv)
    return {'image': sr, 'image_raw': rgb, 'image_depth': depth_img, 'planes': planes}


def make_uniforms(cfg: GenConfig, n: int, seed: int = 4, nrr: Optional[int] = None):
    nrr = cfg.nrr if nrr is None else nrr
    m = nrr * nrr
    u1 = _rand('u1', seed, (n, m, cfg.rendering['depth_resolution'], 1))
    u2 = _rand('u2', seed, (n * m, cfg.rendering['depth_resolution_importance']))
    return u1, u2


# ---------------------------------------------------------------------------------------------------
# Loss glue restatement (Phase A / Phase B)
# ---------------------------------------------------------------------------------------------------


def compute_tv_norm(values):
    """training/coaches/base_coach.py:294-305."""
    v00, v01, v10 = values[:, :-1, :-1], values[:, :-1, 1:], values[:, 1:, :-1]
    return torch.mean(torch.mean((v00 - v01) ** 2 + (v00 - v10) ** 2))


def noise_regularizer(noise_bufs: Sequence[Tensor]):
    """training/projectors/w_projector.py:221-237 (pyramid of shifted auto-correlations)."""
    reg = 0.0
    for v in noise_bufs:
        noise = v[None, None, :, :]
        while True:
            reg = reg + (noise * torch.roll(noise, shifts=1, dims=3)).mean() ** 2
            reg = reg + (noise * torch.roll(noise, shifts=1, dims=2)).mean() ** 2
            if noise.shape[2] <= 8:
                break
            noise = F.avg_pool2d(noise, kernel_size=2)
    return reg


def quaternion_to_rotmat(q):
    """utils/camera_utils.py:201-228 compute_rotation_matrix_from_quaternion (q = [w,x,y,z], normalised first)."""
    q = q / torch.sqrt(torch.clamp((q ** 2).sum(1, keepdim=True), min=1e-8))
    qw, qx, qy, qz = q[:, 0:1], q[:, 1:2], q[:, 2:3], q[:, 3:4]
    xx, yy, zz = qx * qx, qy * qy, qz * qz
    xy, xz, yz = qx * qy, qx * qz, qy * qz
    xw, yw, zw = qx * qw, qy * qw, qz * qw
    r0 = torch.cat((1 - 2 * yy - 2 * zz, 2 * xy - 2 * zw, 2 * xz + 2 * yw), 1)
    r1 = torch.cat((2 * xy + 2 * zw, 1 - 2 * xx - 2 * zz, 2 * yz - 2 * xw), 1)
    r2 = torch.cat((2 * xz - 2 * yw, 2 * yz + 2 * xw, 1 - 2 * xx - 2 * yy), 1)
    return torch.stack((r0, r1, r2), 1)


def rot6d_to_rotmat(x):
    """utils/camera_utils.py:259-273 (Zhou et al. 6-D rotation; the reference offsets every component by 1e-4 first):
    Gram-Schmidt of the two 3-vectors, third axis by cross product, the three axes are the COLUMNS of the result."""
    v = x.reshape(-1, 2, 3) + 1e-4
    e1 = F.normalize(v[:, 0])
    e2 = F.normalize(v[:, 1] - (e1 * v[:, 1]).sum(-1, keepdim=True) * e1)
    e3 = torch.linalg.cross(e1, e2)
    return torch.stack((e1, e2, e3), dim=-1)


def euler_to_rotmat(theta, phi, roll=None, radius: float = 2.7):
    """Rotation block of utils/camera_utils.py:241-257 euler2rot -> :158-188 create_cam2world_matrix_roll: camera on the sphere at
    azimuth theta / polar angle phi looking at the origin (y up), then an in-plane roll.  theta, phi: [B] or [B,1]; returns [B,3,3]."""
    theta, phi = theta.reshape(-1, 1), phi.reshape(-1, 1)
    origin = torch.cat([radius * torch.sin(phi) * torch.cos(math.pi - theta), radius * torch.cos(phi),
                        radius * torch.sin(phi) * torch.sin(math.pi - theta)], 1)
    fwd = -origin / torch.norm(-origin, dim=-1, keepdim=True)
    fwd = fwd / torch.norm(fwd, dim=-1, keepdim=True)
    up0 = torch.tensor([0., 1., 0.]).expand_as(fwd)
    right = torch.linalg.cross(up0, fwd)
    right = -(right / torch.norm(right, dim=-1, keepdim=True))
    up = torch.linalg.cross(fwd, right)
    up = up / torch.norm(up, dim=-1, keepdim=True)
    R0 = torch.stack((right, up, fwd), dim=-1)
    if roll is None:
        return R0
    r = roll.reshape(-1, 1)
    z, o = torch.zeros_like(r), torch.ones_like(r)
    Rz = torch.stack([torch.cat([torch.cos(r), -torch.sin(r), z], 1), torch.cat([torch.sin(r), torch.cos(r), z], 1), torch.cat([z, z, o], 1)], 1)
    return torch.bmm(Rz, R0)


def pose_to_rotmat(pred, mode: str):
    """Dispatch of training/projectors/w_projector.py:147-158: 'quat' (FFHQ default), '6d' (AFHQ), 'euler' (two angles around pi/2)."""
    if mode == 'quat':
        return quaternion_to_rotmat(pred)
    if mode == '6d':
        return rot6d_to_rotmat(pred)
    if mode == 'euler':
        return euler_to_rotmat(math.pi / 2 + pred[:, 0], math.pi / 2 + pred[:, 1], torch.zeros(1, 1))
    raise ValueError(mode)


def l2_loss(a, b):
    """criteria/l2_loss.py:6-8."""
    return F.mse_loss(a, b, reduction='mean')


def psnr_01(img, target):
    """PSNR = -10 log10(MSE) on [0,1]-scaled images (single_id_coach.py:90-94 scaling)."""
    a = (img.clamp(-1, 1) + 1) / 2
    b = (target.clamp(-1, 1) + 1) / 2
    return -10.0 * torch.log10(F.mse_loss(a, b))
