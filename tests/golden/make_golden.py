#!/usr/bin/env python3
"""
Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference, which does not travel).

Imports the reference (cvlab-kaist/3DGAN-Inversion) on CPU -- its pure-PyTorch `_ref` op path -- feeds it
seeded inputs with injected randomness, records inputs + the REFERENCE's outputs/gradients as small .npz
fixtures, and asserts that oracle/eg3d_oracle.py reproduces every one of them (the oracle "pin").

    python tests/golden/make_golden.py            # regenerates tests/golden/*.npz + MANIFEST.json

Nothing under tests/ other than this script reads /root/reference.
"""
import contextlib
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

torch.Tensor.cuda = lambda self, *a, **k: self      # ray_sampler.py:38 calls .cuda() on an unused tensor

from torch_utils.ops import bias_act as ref_bias_act            # noqa: E402
from torch_utils.ops import upfirdn2d as ref_upfirdn2d          # noqa: E402
from torch_utils.ops import conv2d_resample as ref_c2r          # noqa: E402
from torch_utils.ops import filtered_lrelu as ref_flrelu        # noqa: E402
from training import networks_stylegan2 as ref_sg2              # noqa: E402
from training.triplane import TriPlaneGenerator, OSGDecoder     # noqa: E402
from training.volumetric_rendering.renderer import ImportanceRenderer, sample_from_planes, generate_planes  # noqa: E402
from training.volumetric_rendering.ray_sampler import RaySampler  # noqa: E402
from training.volumetric_rendering.ray_marcher import MipRayMarcher2  # noqa: E402
from training.volumetric_rendering import math_utils as ref_math  # noqa: E402

from oracle import eg3d_oracle as O                              # noqa: E402

torch.manual_seed(0)
MANIFEST = {}


def T(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save(name, tol, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: T(v) for k, v in arrays.items()})
    MANIFEST[name] = dict(tol=tol, keys=sorted(arrays.keys()), bytes=os.path.getsize(path))
    print(f'  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB)')


def check(a, b, tol, what):
    a, b = torch.as_tensor(T(a)), torch.as_tensor(T(b))
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item()) if b.numel() else 1.0
    assert err <= tol * scale, f'ORACLE != REFERENCE for {what}: err {err:.3e} (scale {scale:.3e}, tol {tol})'
    return err


def check_adam(a, b, tol, step_bound, what, frac=0.995):
    """Adam-updated state after a few steps: an element whose gradient is at rounding-noise level moves by lr * g / (|g| + eps) -- a
    full step in a direction the reference's own arithmetic does not determine -- so a handful of elements of the 256^2 buffers
    legitimately differ by up to the accumulated step size.  Required: >= frac of the elements within tol, all within step_bound."""
    a, b = torch.as_tensor(T(a)), torch.as_tensor(T(b))
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    scale = max(1.0, b.abs().max().item())
    ok = (err <= tol * scale).float().mean().item()
    assert ok >= frac and err.max().item() <= step_bound, f'ORACLE != REFERENCE for {what}: {100 * ok:.2f} % within {tol}, max {err.max().item():.3e}'
    return err.max().item()


@contextlib.contextmanager
def inject(rand_like=None, rand=None, randn=None):
    """Replay recorded tensors in place of torch.rand_like / torch.rand / torch.randn inside the reference."""
    o_rl, o_r, o_rn = torch.rand_like, torch.rand, torch.randn
    q_rl, q_r, q_rn = list(rand_like or []), list(rand or []), list(randn or [])
    if rand_like is not None:
        torch.rand_like = lambda x, **k: q_rl.pop(0).reshape(x.shape)
    if rand is not None:
        torch.rand = lambda *s, **k: q_r.pop(0)
    if randn is not None:
        torch.randn = lambda *s, **k: q_rn.pop(0)
    try:
        yield
    finally:
        torch.rand_like, torch.rand, torch.randn = o_rl, o_r, o_rn


# ---------------------------------------------------------------------------------------------------
def gen_bias_act():
    print('bias_act')
    out = {}
    g = torch.Generator().manual_seed(11)
    case = 0
    for act in O.ACTIVATIONS:
        for clamp in (None, 0.7):
            for shape, dim in (((2, 5, 4, 3), 1), ((3, 6), 1), ((2, 4, 3, 5), 3)):
                x = (torch.randn(shape, generator=g) * 2).requires_grad_(True)
                b = torch.randn(shape[dim], generator=g).requires_grad_(True)
                dy = torch.randn(shape, generator=g)
                gain = None if case % 3 else 1.3
                alpha = None if case % 2 else 0.1
                y = ref_bias_act.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')
                dx, db = torch.autograd.grad(y, [x, b], dy)
                yo = O.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
                check(yo, y, 1e-6, f'bias_act {act}')
                k = f'c{case}'
                out.update({f'{k}_x': x, f'{k}_b': b, f'{k}_dy': dy, f'{k}_y': y, f'{k}_dx': dx, f'{k}_db': db})
                out[f'{k}_meta'] = np.array([dim, -1 if clamp is None else clamp, -1 if gain is None else gain,
                                             -1 if alpha is None else alpha], dtype=np.float64)
                out[f'{k}_act'] = np.array(act)
                case += 1
    out['ncases'] = np.array(case)
    save('bias_act', 1e-6, **out)


def gen_upfirdn2d():
    print('upfirdn2d')
    g = torch.Generator().manual_seed(12)
    f44 = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    check(O.setup_filter([1, 3, 3, 1]), f44, 1e-7, 'setup_filter')
    f_sep = ref_upfirdn2d.setup_filter([1, 2, 3, 4, 4, 3, 2, 1])          # separable (>= 8 taps)
    check(O.setup_filter([1, 2, 3, 4, 4, 3, 2, 1]), f_sep, 1e-7, 'setup_filter sep')
    f_asym = torch.tensor([[1., 2., 0.5], [0., -1., 3.], [0.25, 1., 1.5], [2., 0.1, -0.3]])   # 4x3 asymmetric
    f_gain = ref_upfirdn2d.setup_filter([1, 3, 3, 1], gain=4)
    check(O.setup_filter([1, 3, 3, 1], gain=4), f_gain, 1e-7, 'setup_filter gain')
    cases = [
        # (shape, filter, up, down, padding, flip, gain)   -- first three are the on-path configs
        ((2, 6, 9, 9), f44, 1, 1, [1, 1, 1, 1], False, 4.0),          # FIR after up-2 transposed conv
        ((2, 6, 8, 8), f44, 2, 1, [2, 1, 2, 1], False, 4.0),          # skip-image upsample2d
        ((1, 3, 8, 8), f44, 1, 1, [2, 2, 2, 2], True, 4.0),           # backward of the first
        ((1, 3, 10, 12), f44, 1, 2, [1, 1, 1, 1], False, 1.0),        # downsample
        ((1, 4, 7, 5), f_sep, 2, 1, [4, 3, 4, 3], False, 4.0),        # separable
        ((2, 3, 9, 8), f_asym, 1, 1, [1, 1, 2, 1], False, 1.0),       # asymmetric non-flipped
        ((2, 3, 9, 8), f_asym, 1, 1, [1, 1, 2, 1], True, 1.0),        # asymmetric flipped
        ((1, 2, 12, 12), f44, 1, 1, [-1, -2, 0, -1], False, 1.0),     # negative padding (crop)
        ((1, 2, 6, 7), f_asym, (2, 1), (1, 2), [2, 1, 3, 2], False, 2.0),   # anisotropic up/down
        ((1, 2, 6, 6), None, 1, 1, 0, False, 1.0),                    # identity
        ((1, 2, 5, 5), f44, 4, 1, [3, 3, 3, 3], False, 16.0),         # up 4
    ]
    out = {}
    for i, (shape, f, up, down, pad, flip, gain) in enumerate(cases):
        x = torch.randn(shape, generator=g).requires_grad_(True)
        y = ref_upfirdn2d.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl='ref')
        dy = torch.randn(y.shape, generator=g)
        dx, = torch.autograd.grad(y, x, dy)
        check(O.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain), y, 1e-6, f'upfirdn2d {i}')
        k = f'c{i}'
        upx, upy = O._xy(up)
        dnx, dny = O._xy(down)
        out.update({f'{k}_x': x, f'{k}_y': y, f'{k}_dy': dy, f'{k}_dx': dx,
                    f'{k}_f': (f if f is not None else np.zeros((0,), np.float32)),
                    f'{k}_meta': np.array([upx, upy, dnx, dny, *O._pad4(pad), int(flip), gain], dtype=np.float64)})
    # wrappers
    x = torch.randn(1, 3, 6, 6, generator=g)
    for nm, fn_r, fn_o in (('upsample2d', ref_upfirdn2d.upsample2d, O.upsample2d),
                           ('downsample2d', ref_upfirdn2d.downsample2d, O.downsample2d),
                           ('filter2d', ref_upfirdn2d.filter2d, O.filter2d)):
        y = fn_r(x, f44, impl='ref')
        check(fn_o(x, f44), y, 1e-6, nm)
        out[f'w_{nm}_y'] = y
    out['w_x'] = x
    out['f44'] = f44
    out['ncases'] = np.array(len(cases))
    save('upfirdn2d', 1e-6, **out)


def gen_filtered_lrelu():
    """filtered_lrelu (torch_utils/ops/filtered_lrelu.py): the reference's `_filtered_lrelu_ref` composition, forward and gradients."""
    print('filtered_lrelu')
    g = torch.Generator().manual_seed(31)
    f12 = ref_upfirdn2d.setup_filter([1, 4, 9, 16, 22, 26, 26, 22, 16, 9, 4, 1][:12], normalize=True)     # 12x12 (outer product)
    f6_1d = torch.tensor([1., 3., 5., 5., 3., 1.]) / 18.0                                                   # separable, 1-D
    f_asym = torch.tensor([[1., 2., 0.5, 0.2], [0., -1., 3., 0.4], [0.25, 1., 1.5, -0.6], [2., 0.1, -0.3, 0.7]]) / 4
    f44 = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    cases = [
        # (shape, fu, fd, up, down, padding, gain, slope, clamp, flip, bias)
        ((2, 3, 8, 8), f12, f12, 2, 2, [5, 5, 5, 5], math.sqrt(2), 0.2, None, False, True),     # StyleGAN3-like up2/down2, 12 taps
        ((1, 4, 6, 7), f6_1d, f6_1d, 2, 1, [3, 2, 3, 2], math.sqrt(2), 0.2, 0.7, False, True),  # separable filters, clamp active
        ((2, 2, 9, 8), f_asym, f44, 1, 2, [2, 1, 1, 2], 1.0, 0.1, None, True, False),           # asymmetric 2-D up filter, flipped, no bias
        ((1, 3, 5, 5), f44, None, 4, 1, [3, 3, 3, 3], 2.0, 0.0, 1.5, False, True),              # up 4, no down filter, relu (slope 0)
        ((1, 2, 10, 10), None, f44, 1, 2, [1, 1, 1, 1], math.sqrt(2), 0.2, None, False, True),  # no up filter
        ((2, 3, 7, 6), None, None, 1, 1, 0, math.sqrt(2), 0.2, 0.5, False, True),               # plain bias + lrelu + clamp
        ((1, 2, 12, 12), f12, f12, 4, 2, [8, 9, 9, 8], math.sqrt(2), 0.2, 256.0, False, True),  # up 4 / down 2, uneven padding
    ]
    out = {}
    for i, (shape, fu, fd, up, down, pad, gain, slope, clamp, flip, has_b) in enumerate(cases):
        x = torch.randn(shape, generator=g).requires_grad_(True)
        b = (torch.randn(shape[1], generator=g) * 0.5).requires_grad_(True) if has_b else None
        y = ref_flrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=pad, gain=gain, slope=slope, clamp=clamp,
                                      flip_filter=flip, impl='ref')
        dy = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad(y, [x] + ([b] if has_b else []), dy)
        yo = O.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=pad, gain=gain, slope=slope, clamp=clamp, flip_filter=flip)
        check(yo, y, 1e-6, f'filtered_lrelu {i}')
        go = torch.autograd.grad(yo, [x] + ([b] if has_b else []), dy)
        for a_, b_ in zip(go, grads):
            check(a_, b_, 1e-5, f'filtered_lrelu {i} grad')
        k = f'c{i}'
        out.update({f'{k}_x': x, f'{k}_y': y, f'{k}_dy': dy, f'{k}_dx': grads[0],
                    f'{k}_b': (b if has_b else np.zeros((0,), np.float32)), f'{k}_db': (grads[1] if has_b else np.zeros((0,), np.float32)),
                    f'{k}_fu': (fu if fu is not None else np.zeros((0,), np.float32)),
                    f'{k}_fd': (fd if fd is not None else np.zeros((0,), np.float32)),
                    f'{k}_meta': np.array([up, down, *O._pad4(pad), gain, slope, -1.0 if clamp is None else clamp, int(flip)], dtype=np.float64)})
    out['ncases'] = np.array(len(cases))
    save('filtered_lrelu', 1e-6, **out)


def gen_conv2d_resample():
    print('conv2d_resample / modulated_conv2d / fc')
    g = torch.Generator().manual_seed(13)
    f44 = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    out = {}
    cases = [
        # (xshape, wshape, up, down, padding, groups, flip_weight)
        ((2, 4, 8, 8), (6, 4, 3, 3), 1, 1, 1, 1, True),      # plain 3x3          (:134)
        ((2, 4, 8, 8), (6, 4, 1, 1), 1, 1, 0, 1, True),      # 1x1                (:134)
        ((2, 4, 8, 8), (6, 4, 3, 3), 2, 1, 1, 1, False),     # up-2 transposed    (:114)
        ((1, 4, 8, 8), (6, 2, 3, 3), 2, 1, 1, 2, False),     # up-2 grouped
        ((1, 4, 8, 8), (6, 4, 3, 3), 1, 2, 1, 1, True),      # down-2 strided     (:108)
        ((1, 4, 8, 8), (6, 4, 1, 1), 1, 2, 0, 1, True),      # 1x1 + down         (:96)
        ((1, 4, 8, 8), (6, 4, 1, 1), 2, 1, 0, 1, True),      # 1x1 + up           (:102)
        ((1, 4, 8, 8), (6, 4, 3, 3), 1, 1, [1, 2, 0, 1], 1, True),   # asymmetric pad -> fallback (:139)
        ((1, 4, 8, 8), (6, 4, 3, 3), 2, 1, 1, 1, True),      # up-2 with flip_weight=True
    ]
    for i, (xs, wsh, up, down, pad, groups, flipw) in enumerate(cases):
        x = torch.randn(xs, generator=g).requires_grad_(True)
        w = torch.randn(wsh, generator=g).requires_grad_(True)
        y = ref_c2r.conv2d_resample(x, w, f=f44, up=up, down=down, padding=pad, groups=groups, flip_weight=flipw)
        dy = torch.randn(y.shape, generator=g)
        dx, dw = torch.autograd.grad(y, [x, w], dy)
        check(O.conv2d_resample(x, w, f=f44, up=up, down=down, padding=pad, groups=groups, flip_weight=flipw), y, 1e-5, f'c2r {i}')
        k = f'c{i}'
        out.update({f'{k}_x': x, f'{k}_w': w, f'{k}_y': y, f'{k}_dy': dy, f'{k}_dx': dx, f'{k}_dw': dw,
                    f'{k}_meta': np.array([up, down, *O._pad4(pad), groups, int(flipw)], dtype=np.int64)})
    out['f44'] = f44
    out['ncases'] = np.array(len(cases))
    save('conv2d_resample', 1e-5, **out)

    # modulated_conv2d
    out = {}
    i = 0
    for up in (1, 2):
        for demod in (True, False):
            for fused in (True, False):
                for with_noise in (True, False):
                    k = 3
                    x = torch.randn(2, 5, 8, 8, generator=g).requires_grad_(True)
                    w = torch.randn(7, 5, k, k, generator=g).requires_grad_(True)
                    s = (torch.randn(2, 5, generator=g) + 1).requires_grad_(True)
                    res = 8 * up
                    nz = torch.randn(res, res, generator=g) * 0.3 if with_noise else None
                    y = ref_sg2.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=f44, demodulate=demod,
                                                 flip_weight=(up == 1), fused_modconv=fused)
                    dy = torch.randn(y.shape, generator=g)
                    dx, dw, ds = torch.autograd.grad(y, [x, w, s], dy)
                    yo = O.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=f44, demodulate=demod,
                                            flip_weight=(up == 1), fused_modconv=fused)
                    check(yo, y, 1e-5, f'modconv {i}')
                    kk = f'c{i}'
                    out.update({f'{kk}_x': x, f'{kk}_w': w, f'{kk}_s': s, f'{kk}_y': y, f'{kk}_dy': dy, f'{kk}_dx': dx,
                                f'{kk}_dw': dw, f'{kk}_ds': ds,
                                f'{kk}_noise': nz if nz is not None else np.zeros((0,), np.float32),
                                f'{kk}_meta': np.array([up, int(demod), int(fused)], dtype=np.int64)})
                    i += 1
    # 1x1 torgb-style
    x = torch.randn(2, 5, 8, 8, generator=g).requires_grad_(True)
    w = torch.randn(3, 5, 1, 1, generator=g).requires_grad_(True)
    s = (torch.randn(2, 5, generator=g) + 1).requires_grad_(True)
    y = ref_sg2.modulated_conv2d(x, w, s, demodulate=False, fused_modconv=True)
    dy = torch.randn(y.shape, generator=g)
    dx, dw, ds = torch.autograd.grad(y, [x, w, s], dy)
    check(O.modulated_conv2d(x, w, s, demodulate=False), y, 1e-5, 'modconv 1x1')
    out.update(dict(t_x=x, t_w=w, t_s=s, t_y=y, t_dy=dy, t_dx=dx, t_dw=dw, t_ds=ds))
    out['f44'] = f44
    out['ncases'] = np.array(i)
    save('modulated_conv2d', 1e-5, **out)

    # FullyConnectedLayer
    out = {}
    for i, (act, lr, binit) in enumerate((('linear', 1.0, 1.0), ('lrelu', 0.01, 0.0), ('linear', 0.5, 0.3))):
        fc = ref_sg2.FullyConnectedLayer(12, 9, activation=act, lr_multiplier=lr, bias_init=binit)
        x = torch.randn(4, 12, generator=g)
        y = fc(x)
        check(O.fully_connected(x, fc.weight, fc.bias, lr, act), y, 1e-6, f'fc {i}')
        out.update({f'c{i}_x': x, f'c{i}_w': fc.weight, f'c{i}_b': fc.bias, f'c{i}_y': y,
                    f'c{i}_lr': np.array(lr), f'c{i}_act': np.array(act)})
    out['ncases'] = np.array(3)
    save('fully_connected', 1e-6, **out)


# ---------------------------------------------------------------------------------------------------
def ref_decoder_from(P, lr_mul=1.0):
    dec = OSGDecoder(32, {'decoder_lr_mul': lr_mul, 'decoder_output_dim': 32})
    sd = {k[len('decoder.'):]: v for k, v in P.items() if k.startswith('decoder.')}
    dec.load_state_dict(sd)
    return dec


def gen_renderer():
    print('renderer pieces')
    g = torch.Generator().manual_seed(14)
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=5)
    opts = dict(cfg.rendering)
    out = {}

    # ray sampler: 3 cameras incl. skew != 0
    c = O.synth_cameras(3, seed=7)
    c2w = c[:, :16].reshape(3, 4, 4).clone().requires_grad_(True)
    K = c[:, 16:].reshape(3, 3, 3).clone()
    K[1, 0, 1] = 0.05
    K[2, 0, 0], K[2, 1, 1], K[2, 0, 2], K[2, 1, 2] = 3.1, 3.7, 0.45, 0.52
    K = K.requires_grad_(True)
    rs = RaySampler()
    ro, rd = rs(c2w, K, 8)
    go, gd = torch.randn(ro.shape, generator=g), torch.randn(rd.shape, generator=g)
    dc2w, dK = torch.autograd.grad([ro, rd], [c2w, K], [go, gd])
    oo, od = O.ray_sampler(c2w, K, 8)
    check(oo, ro, 1e-6, 'ray origins'); check(od, rd, 1e-6, 'ray dirs')
    out.update(dict(rs_c2w=c2w, rs_K=K, rs_o=ro, rs_d=rd, rs_go=go, rs_gd=gd, rs_dc2w=dc2w, rs_dK=dK))
    depth = torch.rand(1, 1, 8, 8, generator=g) + 2
    xyz = rs.calculate_xyz_of_depth(ro[:1], rd[:1], depth[0])
    check(O.calculate_xyz_of_depth(oo[:1], od[:1], depth[0]), xyz, 1e-6, 'xyz_of_depth')
    out.update(dict(rs_depth=depth, rs_xyz=xyz))

    # sample_from_planes: in- and out-of-range coords
    planes = torch.randn(2, 3, 4, 16, 16, generator=g).requires_grad_(True)
    coords = ((torch.rand(2, 50, 3, generator=g) - 0.5) * 1.3).requires_grad_(True)
    feats = sample_from_planes(generate_planes(), planes, coords, padding_mode='zeros', box_warp=1.0)
    gf = torch.randn(feats.shape, generator=g)
    dpl, dco = torch.autograd.grad(feats, [planes, coords], gf)
    check(O.sample_from_planes(planes, coords, 1.0), feats, 1e-6, 'sample_from_planes')
    out.update(dict(sp_planes=planes, sp_coords=coords, sp_feats=feats, sp_gf=gf, sp_dplanes=dpl, sp_dcoords=dco))

    # decoder
    dec = ref_decoder_from(P)
    f_in = torch.randn(2, 3, 40, 32, generator=g).requires_grad_(True)
    o = dec(f_in, None)
    g_rgb, g_sig = torch.randn(o['rgb'].shape, generator=g), torch.randn(o['sigma'].shape, generator=g)
    params = list(dec.parameters())
    grads = torch.autograd.grad([o['rgb'], o['sigma']], [f_in] + params, [g_rgb, g_sig])
    orgb, osig = O.osg_decoder(P, f_in)
    check(orgb, o['rgb'], 1e-6, 'decoder rgb'); check(osig, o['sigma'], 1e-6, 'decoder sigma')
    out.update(dict(dec_in=f_in, dec_rgb=o['rgb'], dec_sigma=o['sigma'], dec_grgb=g_rgb, dec_gsig=g_sig, dec_din=grads[0],
                    dec_dw0=grads[1], dec_db0=grads[2], dec_dw1=grads[3], dec_db1=grads[4],
                    dec_w0=P['decoder.net.0.weight'], dec_b0=P['decoder.net.0.bias'],
                    dec_w1=P['decoder.net.2.weight'], dec_b1=P['decoder.net.2.bias']))

    # ray marcher incl. a zero-density ray (NaN -> inf -> clamp) and white_back
    rm = MipRayMarcher2()
    S = 10
    colors = torch.rand(1, 6, S, 32, generator=g).requires_grad_(True)
    dens = (torch.randn(1, 6, S, 1, generator=g) * 3).detach()
    dens[0, 2] = -80.0                                    # zero-density ray
    dens = dens.requires_grad_(True)
    depths = torch.sort(torch.rand(1, 6, S, 1, generator=g) * 1.05 + 2.25, dim=2)[0]
    depths[0, 4, 3] = depths[0, 4, 4]                     # a tie
    for wb in (False, True):
        o_ = dict(opts, white_back=wb)
        rgb, dep, w = rm(colors, dens, depths, o_)
        orgb, odep, ow = O.ray_march(colors, dens, depths, o_)
        check(orgb, rgb, 1e-6, 'march rgb'); check(odep, dep, 1e-6, 'march depth'); check(ow, w, 1e-6, 'march w')
        out.update({f'rm{int(wb)}_rgb': rgb, f'rm{int(wb)}_depth': dep, f'rm{int(wb)}_w': w})
    g1, g2 = torch.randn(1, 6, 32, generator=g), torch.randn(1, 6, 1, generator=g)
    g2[0, 2] = 0
    rgb, dep, w = rm(colors, dens, depths, opts)
    dcol, dden = torch.autograd.grad([rgb, dep], [colors, dens], [g1, g2])
    out.update(dict(rm_colors=colors, rm_dens=dens, rm_depths=depths, rm_grgb=g1, rm_gdepth=g2, rm_dcolors=dcol, rm_ddens=dden))

    # stratified sampling: fixed / disparity / per-ray tensor limits
    R = ImportanceRenderer()
    ro1 = ro[:2].detach()
    u1 = torch.rand(2, 64, 12, 1, generator=g)
    with inject(rand_like=[u1]):
        d_fix = R.sample_stratified(ro1, 2.25, 3.3, 12, False)
    check(O.sample_stratified(2, 64, 2.25, 3.3, 12, False, u1), d_fix, 1e-6, 'stratified fixed')
    with inject(rand_like=[u1]):
        d_disp = R.sample_stratified(ro1, 2.25, 3.3, 12, True)
    check(O.sample_stratified(2, 64, 2.25, 3.3, 12, True, u1), d_disp, 1e-6, 'stratified disparity')
    rs_t = torch.rand(2, 64, 1, generator=g) * 0.2 + 2.2
    re_t = rs_t + 0.5 + torch.rand(2, 64, 1, generator=g)
    with inject(rand_like=[u1]):
        d_ten = R.sample_stratified(ro1, rs_t, re_t, 12, False)
    check(O.sample_stratified(2, 64, rs_t, re_t, 12, False, u1), d_ten, 1e-6, 'stratified tensor')
    out.update(dict(ss_u1=u1, ss_fixed=d_fix, ss_disp=d_disp, ss_rs=rs_t, ss_re=re_t, ss_tensor=d_ten))
    # the full-size 48-sample linspace (kernel must reproduce torch.linspace rounding)
    u48 = torch.zeros(1, 1, 48, 1)
    with inject(rand_like=[u48]):
        out['ss_lin48'] = R.sample_stratified(ro1[:1, :1], 2.25, 3.3, 48, False)
    check(O.sample_stratified(1, 1, 2.25, 3.3, 48, False, u48), out['ss_lin48'], 0, 'linspace48')

    # importance sampling (+ det), weights with exact zeros and spikes
    z = d_fix
    w = torch.rand(2, 64, 11, 1, generator=g) ** 4
    w[0, :8] = 0.0
    w[1, 3, 5] = 50.0
    u2 = torch.rand(128, 12, generator=g)
    u2[0, 0] = 0.0
    u2[1, 1] = 0.99999994
    with inject(rand=[u2]):
        zi = R.sample_importance(z, w, 12)
    check(O.sample_importance(z, w, 12, u2), zi, 1e-6, 'sample_importance')
    out.update(dict(si_z=z, si_w=w, si_u2=u2, si_out=zi))
    bins = 0.5 * (z.reshape(128, 12)[:, :-1] + z.reshape(128, 12)[:, 1:])
    wts = w.reshape(128, 11)[:, 1:-1]
    pdf_det = R.sample_pdf(bins, wts, 12, det=True)
    check(O.sample_pdf(bins, wts, 12, torch.linspace(0, 1, 12).expand(128, 12)), pdf_det, 1e-6, 'sample_pdf det')
    out['si_det'] = pdf_det

    # unify_samples (with ties)
    d1 = d_fix[:1, :4]
    d2 = zi[:1, :4].clone()
    d2[0, 0, 0] = d1[0, 0, 3]
    c1, c2 = torch.rand(1, 4, 12, 32, generator=g), torch.rand(1, 4, 12, 32, generator=g)
    s1, s2 = torch.randn(1, 4, 12, 1, generator=g), torch.randn(1, 4, 12, 1, generator=g)
    ud, uc, us = R.unify_samples(d1, c1, s1, d2, c2, s2)
    od_, oc_, os_ = O.unify_samples(d1, c1, s1, d2, c2, s2)
    check(od_, ud, 0, 'unify depths')
    out.update(dict(us_d1=d1, us_d2=d2, us_c1=c1, us_c2=c2, us_s1=s1, us_s2=s2, us_d=ud, us_c=uc, us_s=us))

    # box limits incl. misses
    ob = torch.tensor([[0., 0., 2.7], [0., 0., 2.7], [2.7, 0., 0.], [0., 3., 0.]])[None]
    db = torch.nn.functional.normalize(torch.tensor([[0., 0.05, -1.], [0., 0.6, -1.], [-1., 0.1, 0.1], [0.2, -1., 0.1]]), dim=1)[None]
    t0, t1 = ref_math.get_ray_limits_box(ob, db, box_side_length=1.0)
    q0, q1 = O.get_ray_limits_box(ob, db, 1.0)
    check(q0, t0, 1e-6, 'box tmin'); check(q1, t1, 1e-6, 'box tmax')
    out.update(dict(box_o=ob, box_d=db, box_tmin=t0, box_tmax=t1))

    # full ImportanceRenderer.forward on small planes, with gradients
    planes = (torch.randn(2, 3, 32, 16, 16, generator=g) * 0.7).requires_grad_(True)
    cam = O.synth_cameras(2, seed=9)
    c2w = cam[:, :16].reshape(2, 4, 4).clone().requires_grad_(True)
    K = cam[:, 16:].reshape(2, 3, 3)
    ro, rd = rs(c2w, K, 6)
    u1 = torch.rand(2, 36, 12, 1, generator=g)
    u2 = torch.rand(72, 12, generator=g)
    with inject(rand_like=[u1], rand=[u2]):
        rgb, dep, wsum = R(planes, dec, ro, rd, opts)
    g_rgb, g_dep = torch.randn(rgb.shape, generator=g), torch.randn(dep.shape, generator=g)
    grads = torch.autograd.grad([rgb, dep], [planes, c2w] + params, [g_rgb, g_dep])
    oro, ord_ = O.ray_sampler(c2w, K, 6)
    orgb, odep, owsum = O.render(P, planes, oro, ord_, opts, u1, u2)
    e1 = check(orgb, rgb, 2e-6, 'render rgb'); e2 = check(odep, dep, 2e-6, 'render depth'); check(owsum, wsum, 2e-6, 'render wsum')
    og = torch.autograd.grad([orgb, odep], [planes, c2w], [g_rgb, g_dep])
    check(og[0], grads[0], 1e-5, 'render dplanes'); check(og[1], grads[1], 1e-5, 'render dc2w')
    print(f'    render err rgb {e1:.2e} depth {e2:.2e}')
    out.update(dict(rn_planes=planes, rn_c2w=c2w, rn_K=K, rn_u1=u1, rn_u2=u2, rn_rgb=rgb, rn_depth=dep, rn_wsum=wsum,
                    rn_grgb=g_rgb, rn_gdepth=g_dep, rn_dplanes=grads[0], rn_dc2w=grads[1], rn_dw0=grads[2], rn_db0=grads[3],
                    rn_dw1=grads[4], rn_db1=grads[5]))
    # 'auto' ray limits branch (forward only)
    o_auto = dict(opts, ray_start='auto', ray_end='auto')
    with inject(rand_like=[u1], rand=[u2]):
        rgb_a, dep_a, _ = R(planes, dec, ro, rd, o_auto)
    orgb_a, odep_a, _ = O.render(P, planes, oro, ord_, o_auto, u1, u2)
    check(orgb_a, rgb_a, 2e-6, 'render auto rgb'); check(odep_a, dep_a, 2e-6, 'render auto depth')
    out.update(dict(rn_auto_rgb=rgb_a, rn_auto_depth=dep_a))
    save('renderer', 2e-6, **out)


# ---------------------------------------------------------------------------------------------------
class RefComposite(torch.nn.Module):
    """The reference's own classes composed exactly as TriPlaneGenerator does (triplane.py:36-45,53-90), but with
    free plane / SR sizes (the reference class hard-codes 256 / 512)."""

    def __init__(self, cfg, P):
        super().__init__()
        self.cfg = cfg
        self.backbone = ref_sg2.Generator(cfg.z_dim, cfg.c_dim, cfg.w_dim, img_resolution=cfg.plane_res,
                                          img_channels=cfg.plane_channels, mapping_kwargs={'num_layers': cfg.mapping_layers},
                                          channel_base=cfg.channel_base, channel_max=cfg.channel_max,
                                          fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None)
        c0, c1 = cfg.sr_channels
        self.block0 = ref_sg2.SynthesisBlock(32, c0, w_dim=cfg.w_dim, resolution=cfg.sr_in_res * 2, img_channels=3, is_last=False,
                                             use_fp16=True, conv_clamp=256, fused_modconv_default='inference_only')
        self.block1 = ref_sg2.SynthesisBlock(c0, c1, w_dim=cfg.w_dim, resolution=cfg.sr_in_res * 4, img_channels=3, is_last=True,
                                             use_fp16=True, conv_clamp=256, fused_modconv_default='inference_only')
        self.decoder = ref_decoder_from(P)
        self.renderer = ImportanceRenderer()
        self.ray_sampler = RaySampler()
        sd = {}
        for k, v in P.items():
            if k.startswith('backbone.'):
                sd[k] = v
            elif k.startswith('superresolution.'):
                sd[k[len('superresolution.'):]] = v
        missing, unexpected = self.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(m.startswith('decoder.') for m in missing), missing
        self.eval().float()

    def synthesis(self, ws, c, u1, u2, noise_mode='const', randn=None, fused_modconv=None):
        cfg = self.cfg
        c2w = c[:, :16].view(-1, 4, 4)
        K = c[:, 16:25].view(-1, 3, 3)
        ro, rd = self.ray_sampler(c2w, K, cfg.nrr)
        with inject(randn=randn):
            planes = self.backbone.synthesis(ws, noise_mode=noise_mode, force_fp32=True, fused_modconv=fused_modconv)
        pl = planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])
        with inject(rand_like=[u1], rand=[u2]):
            feat, depth, _ = self.renderer(pl, self.decoder, ro, rd, cfg.rendering)
        n = ws.shape[0]
        feat_img = feat.permute(0, 2, 1).reshape(n, 32, cfg.nrr, cfg.nrr).contiguous()
        depth_img = depth.permute(0, 2, 1).reshape(n, 1, cfg.nrr, cfg.nrr)
        rgb = feat_img[:, :3]
        ws3 = ws[:, -1:, :].repeat(1, 3, 1)                                  # superresolution.py:280
        x, img = self.block0(feat_img, rgb, ws3, noise_mode='none', force_fp32=True, fused_modconv=fused_modconv)
        x, img = self.block1(x, img, ws3, noise_mode='none', force_fp32=True, fused_modconv=fused_modconv)
        return {'image': img, 'image_raw': rgb, 'image_depth': depth_img, 'planes': planes}


def gen_graph_small():
    print('small generator graph')
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=0)
    G = RefComposite(cfg, P)
    n = 2
    ws = O.synth_ws(cfg, n, seed=1, wplus=True).requires_grad_(True)
    c = O.synth_cameras(n, seed=2).requires_grad_(True)
    u1, u2 = O.make_uniforms(cfg, n, seed=4)
    out = {}
    # mapping network
    z = O._randn('z', 3, (n, cfg.z_dim))
    wmap = G.backbone.mapping(z, c.detach() * 1.0, truncation_psi=0.7, truncation_cutoff=5)
    check(O.mapping(P, cfg, z, c.detach(), 0.7, 5), wmap, 1e-5, 'mapping')
    out.update(dict(map_z=z, map_out=wmap))

    for mode in ('const', 'random'):
        randn = None
        noises = None
        if mode == 'random':
            # one randn per backbone SynthesisLayer, in call order (networks_stylegan2.py:318-319)
            noises, randn = {}, []
            for r in cfg.block_resolutions:
                for conv in (['conv1'] if r == 4 else ['conv0', 'conv1']):
                    nm = f'backbone.synthesis.b{r}.{conv}'
                    t = O._randn('noise.' + nm, 6, (n, 1, r, r))
                    noises[nm] = t
                    randn.append(t)
        tr_params = [G.backbone.synthesis.b8.conv0.weight, G.backbone.synthesis.b16.torgb.weight, G.block1.conv1.weight,
                     G.decoder.net[0].weight, G.backbone.synthesis.b32.conv1.noise_strength, G.backbone.synthesis.b16.conv0.bias,
                     G.block0.conv0.affine.weight]
        tr_names = ['backbone.synthesis.b8.conv0.weight', 'backbone.synthesis.b16.torgb.weight',
                    'superresolution.block1.conv1.weight', 'decoder.net.0.weight',
                    'backbone.synthesis.b32.conv1.noise_strength', 'backbone.synthesis.b16.conv0.bias',
                    'superresolution.block0.conv0.affine.weight']
        o = G.synthesis(ws, c, u1, u2, noise_mode=mode, randn=randn)
        g_img = O._randn('g_img', 8, o['image'].shape)
        g_raw = O._randn('g_raw', 8, o['image_raw'].shape)
        g_dep = O._randn('g_dep', 8, o['image_depth'].shape)
        grads = torch.autograd.grad([o['image'], o['image_raw'], o['image_depth']], [ws, c] + tr_params, [g_img, g_raw, g_dep])
        Pg = {k: (v.clone().requires_grad_(True) if k in tr_names else v) for k, v in P.items()}
        oo = O.synthesis(Pg, cfg, ws, c, u1, u2, noise_mode=mode, noises=noises)
        for k in ('image', 'image_raw', 'image_depth', 'planes'):
            e = check(oo[k], o[k], 2e-5, f'graph[{mode}] {k}')
            print(f'    [{mode}] {k}: max err {e:.2e}')
        ograds = torch.autograd.grad([oo['image'], oo['image_raw'], oo['image_depth']], [ws, c] + [Pg[k] for k in tr_names],
                                     [g_img, g_raw, g_dep])
        for nm, a, b in zip(['ws', 'c'] + tr_names, ograds, grads):
            e = check(a, b, 2e-4, f'graph[{mode}] grad {nm}')
        # non-fused formulation is interchangeable
        onf = O.synthesis(P, cfg, ws, c, u1, u2, noise_mode=mode, noises=noises, fused_modconv=False)
        check(onf['image'], o['image'], 1e-4, 'non-fused image')
        m = mode[0]
        out.update({f'{m}_image': o['image'], f'{m}_raw': o['image_raw'], f'{m}_depth': o['image_depth'],
                    f'{m}_planes': o['planes'], f'{m}_dws': grads[0], f'{m}_dc': grads[1]})
        for nm, gval in zip(tr_names, grads[2:]):
            out[f'{m}_d.{nm}'] = gval
    out.update(dict(ws=ws, c=c, g_img=g_img, g_raw=g_raw, g_dep=g_dep))
    # noise_const gradient (Phase A optimises them): w.r.t. one buffer
    save('graph_small', 2e-5, **out)


FULL_WGRAD_KEYS = ['backbone.synthesis.b4.conv1.weight', 'backbone.synthesis.b32.conv1.weight', 'backbone.synthesis.b64.conv0.weight',
                   'backbone.synthesis.b128.conv0.weight', 'backbone.synthesis.b128.conv1.weight', 'backbone.synthesis.b256.conv0.weight',
                   'backbone.synthesis.b256.conv1.weight', 'backbone.synthesis.b256.torgb.weight', 'backbone.synthesis.b64.torgb.bias',
                   'backbone.synthesis.b64.conv1.noise_strength', 'backbone.synthesis.b256.conv1.noise_strength',
                   'backbone.synthesis.b128.conv1.bias', 'backbone.synthesis.b64.conv1.affine.weight', 'backbone.synthesis.b256.conv0.affine.bias',
                   'superresolution.block0.conv0.weight', 'superresolution.block0.conv1.weight', 'superresolution.block1.conv0.weight',
                   'superresolution.block1.conv1.weight', 'superresolution.block1.torgb.weight', 'superresolution.block1.conv1.affine.weight',
                   'decoder.net.0.weight', 'decoder.net.0.bias', 'decoder.net.2.weight', 'decoder.net.2.bias']
FULL_NOISE_KEYS = ['backbone.synthesis.b256.conv1.noise_const', 'backbone.synthesis.b32.conv0.noise_const']


def gen_graph_full():
    """Full-size ffhqrebalanced512-128-shaped generator: the reference's own TriPlaneGenerator class.
    Only probe samples + statistics are stored (weights come from the deterministic generator)."""
    print('full-size generator (reference TriPlaneGenerator) -- takes a minute')
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    rk = dict(cfg.rendering, superresolution_module='training.superresolution.SuperresolutionHybrid8XDC')
    G = TriPlaneGenerator(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs={'num_layers': 2},
                          rendering_kwargs=rk, channel_base=32768, channel_max=512, fused_modconv_default='inference_only',
                          num_fp16_res=0, sr_num_fp16_res=4,
                          sr_kwargs={'channel_base': 32768, 'channel_max': 512, 'fused_modconv_default': 'inference_only'},
                          conv_clamp=None)
    G.neural_rendering_resolution = 128
    missing, unexpected = G.load_state_dict(P, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    G.eval().float()
    ws = O.synth_ws(cfg, 1, seed=1).requires_grad_(True)
    c = O.synth_cameras(1, seed=2).requires_grad_(True)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    nbufs = [dict(G.named_buffers())[k].requires_grad_(True) for k in FULL_NOISE_KEYS]
    with inject(rand_like=[u1], rand=[u2]):
        o = G.synthesis(ws, c, noise_mode='const', force_fp32=True)
    probe = torch.Generator().manual_seed(99)
    idx_img = torch.randint(0, 3 * 512 * 512, (4096,), generator=probe)
    idx_raw = torch.randint(0, 3 * 128 * 128, (2048,), generator=probe)
    idx_dep = torch.randint(0, 128 * 128, (2048,), generator=probe)
    g_img = O._randn('gf_img', 8, o['image'].shape) / (3 * 512 * 512)
    g_dep = O._randn('gf_dep', 8, o['image_depth'].shape) / (128 * 128)
    # Phase B at full size (base_coach.py:96-99: Adam over every generator weight): weight-gradient probes from the reference class
    # itself -- 256 fixed indices + L2 norm + max|g| per tensor -- plus the gradient of two noise_const buffers (Phase A leaves)
    named = dict(G.named_parameters())
    wg = [named[k] for k in FULL_WGRAD_KEYS]
    grads = torch.autograd.grad([o['image'], o['image_depth']], [ws, c] + wg + nbufs, [g_img, g_dep])
    dws, dc = grads[:2]
    wprobe = {}
    for k, gval in zip(FULL_WGRAD_KEYS + FULL_NOISE_KEYS, grads[2:]):
        flat = gval.detach().flatten()
        idx = torch.randint(0, flat.numel(), (min(256, flat.numel()),), generator=probe)
        wprobe['wg_idx.' + k] = idx
        wprobe['wg_val.' + k] = flat[idx]
        wprobe['wg_stat.' + k] = np.array([flat.norm().item(), flat.abs().max().item()])
        print(f'    d {k}: norm {flat.norm().item():.3e}  max {flat.abs().max().item():.3e}')
    with torch.no_grad():
        oo = O.synthesis(P, cfg, ws.detach(), c.detach(), u1, u2, noise_mode='const')
    for k in ('image', 'image_raw', 'image_depth'):
        e = check(oo[k], o[k], 1e-4, f'full {k}')
        mse = torch.mean((oo[k] - o[k]) ** 2).item()
        print(f'    {k}: max err {e:.2e}, psnr vs ref {(-10*math.log10(max(mse,1e-30)/4.0)):.1f} dB')
    stats = lambda t: np.array([t.mean().item(), t.abs().mean().item(), t.min().item(), t.max().item()])
    save('graph_full', 1e-4, ws=ws, c=c, idx_img=idx_img, idx_raw=idx_raw, idx_dep=idx_dep,
         img_probe=o['image'].flatten()[idx_img], raw_probe=o['image_raw'].flatten()[idx_raw],
         dep_probe=o['image_depth'].flatten()[idx_dep], img_stats=stats(o['image']), raw_stats=stats(o['image_raw']),
         dep_stats=stats(o['image_depth']), dws=dws, dc=dc, **wprobe)


def cond_loss_cotangents(img, dep):
    """d/d(image, depth) of a smooth objective -- L = mean((avg_pool2(image) - target)^2) + 0.1 mean(depth^2), target a fixed low-pass image in [-1, 1] --
    the shape of the image term of w_projector.py:232-249 at 256^2.  Unlike the white-noise cotangents of graph_full, d c under it is well conditioned
    (VERDICT r5 item 6).  Shared by this generator and tests/test_gpu_generator.py."""
    import torch.nn.functional as F
    low = O._randn('cond_target', 11, (1, 3, 16, 16))
    target = torch.tanh(F.interpolate(low, size=(256, 256), mode='bilinear', align_corners=False)).to(img.device)
    img = img.detach().requires_grad_(True)
    dep = dep.detach().requires_grad_(True)
    L = (F.avg_pool2d(img, 2) - target).square().mean() + 0.1 * dep.square().mean()
    g_img, g_dep = torch.autograd.grad(L, [img, dep])
    return g_img, g_dep


def gen_graph_full_cond():
    """Full-size generator, reference class, under (i) the N(0,1) synthetic weights and (ii) the heavy-tailed ones (O.heavy_tailed_params: log-normal
    per-channel gains, x100 const channels, x10 affine-bias entries, noise_strength up to 1) -- image / raw / depth probes, output ranges, and
    d ws, d c under the conditioned loss cotangent.  Pins the f16x3 range normalisation at GRAPH level on trained-checkpoint statistics."""
    cfg = O.full_config()
    rk = dict(cfg.rendering, superresolution_module='training.superresolution.SuperresolutionHybrid8XDC')
    arrays = {}
    for tag in ('plain', 'heavy'):
        print(f'full-size generator, conditioned cotangent, weights: {tag} -- takes a minute')
        P = O.synth_params(cfg, seed=0)
        if tag == 'heavy':
            P = O.heavy_tailed_params(P, seed=0)
        G = TriPlaneGenerator(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs={'num_layers': 2},
                              rendering_kwargs=rk, channel_base=32768, channel_max=512, fused_modconv_default='inference_only',
                              num_fp16_res=0, sr_num_fp16_res=4,
                              sr_kwargs={'channel_base': 32768, 'channel_max': 512, 'fused_modconv_default': 'inference_only'},
                              conv_clamp=None)
        G.neural_rendering_resolution = 128
        missing, unexpected = G.load_state_dict(P, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        G.eval().float()
        ws = O.synth_ws(cfg, 1, seed=1).requires_grad_(True)
        c = O.synth_cameras(1, seed=2).requires_grad_(True)
        u1, u2 = O.make_uniforms(cfg, 1, seed=4)
        with inject(rand_like=[u1], rand=[u2]):
            o = G.synthesis(ws, c, noise_mode='const', force_fp32=True)
        assert all(torch.isfinite(o[k]).all() for k in ('image', 'image_raw', 'image_depth'))
        g_img, g_dep = cond_loss_cotangents(o['image'], o['image_depth'])
        dws, dc = torch.autograd.grad([o['image'], o['image_depth']], [ws, c], [g_img, g_dep])
        assert torch.isfinite(dws).all() and torch.isfinite(dc).all(), 'the reference\'s own gradient is not finite under these weights'
        probe = torch.Generator().manual_seed(98)
        idx_img = torch.randint(0, 3 * 512 * 512, (4096,), generator=probe)
        idx_raw = torch.randint(0, 3 * 128 * 128, (2048,), generator=probe)
        idx_dep = torch.randint(0, 128 * 128, (2048,), generator=probe)
        with torch.no_grad():
            oo = O.synthesis(P, cfg, ws.detach(), c.detach(), u1, u2, noise_mode='const')
        for k in ('image', 'image_raw', 'image_depth'):
            rng = float(o[k].max() - o[k].min())
            e = check(oo[k] / rng, o[k] / rng, 1e-4, f'full cond {tag} {k} (relative to the output range {rng:.3g})')
        stats = lambda t: np.array([t.mean().item(), t.abs().mean().item(), t.min().item(), t.max().item()])
        print(f'    image range [{o["image"].min().item():.3g}, {o["image"].max().item():.3g}], |d ws| max {dws.abs().max().item():.3e}, |d c| max {dc.abs().max().item():.3e}')
        arrays.update({f'{tag}.idx_img': idx_img, f'{tag}.idx_raw': idx_raw, f'{tag}.idx_dep': idx_dep,
                       f'{tag}.img_probe': o['image'].flatten()[idx_img], f'{tag}.raw_probe': o['image_raw'].flatten()[idx_raw],
                       f'{tag}.dep_probe': o['image_depth'].flatten()[idx_dep], f'{tag}.img_stats': stats(o['image']), f'{tag}.raw_stats': stats(o['image_raw']),
                       f'{tag}.dep_stats': stats(o['image_depth']), f'{tag}.dws': dws, f'{tag}.dc': dc})
        if tag == 'plain':
            arrays['ws'], arrays['c'] = ws, c
        del G, o
    save('graph_full_cond', 1e-4, **arrays)


# ---------------------------------------------------------------------------------------------------
# Loop-level pins.  training/projectors/w_projector.py, training/coaches/*.py import wandb / lpips / torchvision / mrcfile
# (absent here), so their loop bodies are lifted out of the source by AST and executed UNMODIFIED in a namespace that holds the
# reference's own generator classes / RaySampler / calc_warping_loss / camera utilities plus stub perceptual networks.
# ---------------------------------------------------------------------------------------------------
import ast          # noqa: E402
import types        # noqa: E402


def _parse(rel):
    return ast.parse(open(os.path.join(REF, rel)).read())


def _func(tree, name, cls=None):
    body = tree.body
    if cls is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    return [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]


def _exec_nodes(nodes, ns, tag):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, f'<lifted {tag}>', 'exec'), ns)
    return ns


def _lifted_loop(for_node, n_iter_expr, trace_src):
    """The reference `for` statement with its iterable replaced by range(<n_iter_expr>) and one trace statement appended."""
    node = ast.parse(ast.unparse(for_node)).body[0]
    node.iter = ast.parse(f'range({n_iter_expr})').body[0].value
    node.body = node.body + ast.parse(trace_src).body
    return node


class _Lambda(torch.nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)


def _stub_torch_vgg(fw):
    """22 children like torchvision vgg16().features[:22] as get_features (warping_loss.py:74-110) walks them; children 0-5 are the
    oracle's two-stage stub feature map, the rest identities, so layers='14' returns that map."""
    mods = []
    pad4 = lambda x: torch.cat([x, x.new_zeros(*x.shape[:-3], 4 - x.shape[-3], *x.shape[-2:])], -3) if x.shape[-3] < 4 else x     # noqa: E731
    for wt in fw[:2]:
        mods += [_Lambda(lambda x, wt=wt: torch.nn.functional.conv2d(pad4(x), wt, padding=1)), _Lambda(lambda x: O.bias_act(x, None, act='lrelu')),
                 torch.nn.AvgPool2d(2)]
    mods += [torch.nn.Identity() for _ in range(22 - len(mods))]
    return torch.nn.Sequential(*mods)


class _GSteps:
    """G as the lifted loops see it: .synthesis(ws, c, **kw) of the reference classes with the step's recorded uniforms replayed
    (calls_per_step consecutive calls share one (u1, u2): the optimised view and the canonical view of calc_warping_loss)."""

    def __init__(self, ref_composite, uniforms, calls_per_step, randn=None):
        self.R, self.uniforms, self.cps, self.calls, self.randn = ref_composite, uniforms, calls_per_step, 0, randn
        self.backbone = ref_composite.backbone

    def synthesis(self, ws, c, noise_mode='random', force_fp32=False, **kw):
        k = self.calls // self.cps
        self.calls += 1
        u1, u2 = self.uniforms[k]
        return self.R.synthesis(ws, c, u1, u2, noise_mode=noise_mode, randn=None if self.randn is None else self.randn[k])

    def parameters(self):
        return self.R.parameters()


def _pad3(img):
    return torch.cat([img, img.new_zeros(img.shape[0], 1, *img.shape[2:])], 1) if img.dim() == 4 else \
        torch.cat([img, img.new_zeros(1, *img.shape[1:])], 0)


def gen_loss_glue():
    print('loss glue (reference functions; restatement-free)')
    from utils import camera_utils as ref_cam
    from training.warping_loss import LinePlaneCollision
    from training.explainability_network.loss_functions import photometric_reconstruction_loss
    from oracle import inversion_oracle as IO
    g = torch.Generator().manual_seed(15)
    out = {}
    # rotation parametrisations (utils/camera_utils.py:201-228, 259-273, 241-257)
    q = torch.randn(3, 4, generator=g)
    Rq = ref_cam.compute_rotation_matrix_from_quaternion(q)
    check(O.quaternion_to_rotmat(q), Rq, 1e-6, 'quat->R')
    x6 = torch.randn(4, 6, generator=g)          # (not 3: the reference's dim-less torch.cross would pick the batch axis)
    R6 = ref_cam.rot6d_to_rotmat(x6)
    check(O.rot6d_to_rotmat(x6), R6, 1e-6, '6d->R')
    ang = torch.randn(3, 2, generator=g) * 0.3
    Re = torch.cat([ref_cam.euler2rot(math.pi / 2 + a[0:1], math.pi / 2 + a[1:2], torch.zeros(1, 1), batch_size=1).reshape(-1, 4, 4)[:, :3, :3]
                    for a in ang])
    check(torch.cat([O.pose_to_rotmat(a[None], 'euler') for a in ang]), Re, 1e-6, 'euler->R')
    roll = torch.tensor([[0.3]])
    Rr = ref_cam.euler2rot(torch.tensor([1.2]), torch.tensor([1.9]), roll, batch_size=1).reshape(-1, 4, 4)
    check(O.euler_to_rotmat(torch.tensor([1.2]), torch.tensor([1.9]), roll), Rr[:, :3, :3], 1e-6, 'euler+roll->R')
    out.update(dict(q=q, R=Rq, x6=x6, R6=R6, ang=ang, Re=Re, roll_R=Rr[:, :3, :3]))
    # lookat
    origin = torch.tensor([[0.3, 0.5, 2.6]])
    fwd = ref_math.normalize_vecs(-origin)
    m = ref_cam.create_cam2world_matrix(fwd, origin)
    check(O.lookat_cam2world(origin[0], torch.zeros(3)), m[0], 1e-6, 'lookat')
    out.update(dict(lookat_origin=origin, lookat=m))
    # pose -> extrinsic / camera block and the noise regulariser: the reference's own statements out of the body of the optimisation
    # loop (w_projector.py:147-172 and :221-237)
    proj = _func(_parse('training/projectors/w_projector.py'), 'project')
    loop = [n for n in proj.body if isinstance(n, ast.For)][-1]
    first_reg = [i for i, n in enumerate(loop.body) if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]) == 'reg_loss'][0]
    reg_nodes = loop.body[first_reg:first_reg + 3]
    assert [type(n).__name__ for n in reg_nodes] == ['Assign', 'For', 'For'], reg_nodes
    last_pose = [i for i, n in enumerate(loop.body) if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]) == 'pred_cam'][0]
    pose_nodes = loop.body[:last_pose + 1]
    from configs import global_config as gc
    intrinsic = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).unsqueeze(0)
    saved = (gc.use_quaternions, gc.use_6d)
    try:
        for mode, pred in (('quat', torch.randn(1, 4, generator=g)), ('6d', torch.randn(1, 6, generator=g)), ('euler', torch.randn(1, 2, generator=g) * 0.3)):
            gc.use_quaternions, gc.use_6d = mode == 'quat', mode == '6d'
            tr = (torch.randn(1, 3, generator=g) * 0.1).requires_grad_(True)
            pr = pred.clone().requires_grad_(True)
            ns = dict(torch=torch, math=math, global_config=gc, cam_predictor=lambda x: pr, target_images=None, radius=2.7,
                      compute_rotation_matrix_from_quaternion=ref_cam.compute_rotation_matrix_from_quaternion,
                      rot6d_to_rotmat=ref_cam.rot6d_to_rotmat, euler2rot=ref_cam.euler2rot, translation_opt=tr, intrinsic=intrinsic)
            _exec_nodes(pose_nodes, ns, 'pose block')
            gcam = torch.randn(1, 25, generator=g)
            d_pr, d_tr = torch.autograd.grad(ns['pred_cam'], [pr, tr], gcam)
            pr2, tr2 = pred.clone().requires_grad_(True), tr.detach().clone().requires_grad_(True)
            ext, cam = IO.pose_to_cam(O.pose_to_rotmat(pr2, mode), tr2, intrinsic, 2.7)
            check(ext, ns['pred_ext'], 1e-6, f'pose->ext {mode}')
            check(cam, ns['pred_cam'], 1e-6, f'pose->cam {mode}')
            e_pr, e_tr = torch.autograd.grad(cam, [pr2, tr2], gcam)
            check(e_pr, d_pr, 2e-5, f'd cam / d pose {mode}')
            check(e_tr, d_tr, 2e-5, f'd cam / d translation {mode}')
            out.update({f'pose_{mode}_pred': pred, f'pose_{mode}_tr': tr, f'pose_{mode}_cam': ns['pred_cam'], f'pose_{mode}_gcam': gcam,
                        f'pose_{mode}_dpred': d_pr, f'pose_{mode}_dtr': d_tr})
    finally:
        gc.use_quaternions, gc.use_6d = saved
    bufs = [torch.randn(r, r, generator=g) for r in (4, 8, 16, 32)]
    bufs2 = [torch.randn(r, r, generator=g) for r in (16, 64)]
    ns = dict(torch=torch, F=torch.nn.functional, noise_bufs={str(i): b for i, b in enumerate(bufs)}, noise_bufs2={str(i): b for i, b in enumerate(bufs2)})
    _exec_nodes(reg_nodes, ns, 'noise regulariser')
    reg = ns['reg_loss']
    check(O.noise_regularizer(bufs + bufs2), reg, 1e-6, 'noise reg')
    # depth TV: the reference function itself (base_coach.py:294-305), lifted out of its un-importable module
    ns = _exec_nodes([_func(_parse('training/coaches/base_coach.py'), 'compute_tv_norm')], dict(torch=torch), 'compute_tv_norm')
    d = torch.rand(1, 16, 16, generator=g)
    tv = ns['compute_tv_norm'](d)
    check(O.compute_tv_norm(d), tv, 1e-7, 'tv')
    out.update(dict(tv_in=d, tv=tv, reg=reg, **{f'reg_buf{i}': b for i, b in enumerate(bufs + bufs2)}))
    # line-plane intersection + masked photometric loss (warping_loss.py:58-72, loss_functions.py:9-19)
    n_ = 50
    pn, pp, rd, rp = (torch.randn(n_, 3, generator=g) for _ in range(4))
    psi = LinePlaneCollision(pn, pp, rd, rp)
    check(IO.line_plane_collision(pn, pp, rd, rp), psi, 1e-5, 'LinePlaneCollision')
    a, b_, mk = torch.randn(1, 5, 6, 6, generator=g), torch.randn(5, 6, 6, generator=g), torch.rand(1, 1, 6, 6, generator=g)
    ph = photometric_reconstruction_loss(a, b_, mk)
    check(((a - b_) * mk).abs().mean(), ph, 1e-7, 'photometric')
    out.update(dict(lpc_n=pn, lpc_p=pp, lpc_d=rd, lpc_o=rp, lpc_out=psi))
    save('loss_glue', 1e-6, **out)


def gen_projector_loop():
    """Phase A: the reference's optimisation-loop body (w_projector.py:145-270) executed as is -- pose chain, synthesis, calc_warping_loss
    (the reference's own function, imported), LPIPS-feature distance, noise regulariser, optimiser order, noise renormalisation -- for the
    quaternion / 6-D / Euler pose modes; ProjectorOracle must reproduce every recorded step."""
    print('projector loop (reference loop body, lifted)')
    from utils import camera_utils as ref_cam
    from training.warping_loss import calc_warping_loss
    from configs import global_config as gc, hyperparameters as hp
    from oracle import inversion_oracle as IO
    cfg = IO.pin_config()
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)                          # [3,H,W] in [-1,1]  (w_projector.project's `target`)
    PROJ_STEPS, PROJ_PREHEAT = IO.PIN_PROJ_STEPS, IO.PIN_PROJ_PREHEAT
    fw = IO.stub_feature_weights()
    loop = [n for n in _func(_parse('training/projectors/w_projector.py'), 'project').body if isinstance(n, ast.For)][-1]
    node = _lifted_loop(loop, 'num_steps', "_trace.append((float(loss), float(dist), float(reg_loss), float(warp_loss), _psnr(pred_dict['image']))); _gtrace.append((translation_opt.grad.clone(), cam_predictor.base.grad.clone()))")
    out = dict(target_probe=target.flatten()[::37].clone())       # the target itself is regenerated by the tests (oracle render, seed 31)
    saved = (gc.use_quaternions, gc.use_6d, gc.visualize_opt_process, gc.visualize_warp_process, hp.cam_preheat_steps)
    o_randn_like = torch.randn_like
    try:
        gc.visualize_opt_process = gc.visualize_warp_process = False
        hp.cam_preheat_steps = PROJ_PREHEAT
        for mode in ('quat', '6d', 'euler'):
            gc.use_quaternions, gc.use_6d = mode == 'quat', mode == '6d'
            G = RefComposite(cfg, P).requires_grad_(False)
            pin = IO.pin_projector_inputs(cfg, P, mode)
            uniforms, wns, init_noise, w0, base = pin['uniforms'], pin['wns'], pin['init_noise'], pin['w0'], pin['pose_base']
            cam_predictor = IO.StubPoseNet(base, seed=7)
            Gs = _GSteps(G, uniforms, calls_per_step=2)
            noise_bufs = {n: b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n}
            noise_bufs2 = {f'{blk}.{n}': b for blk in ('block0', 'block1') for n, b in getattr(G, blk).named_buffers() if 'noise_const' in n}
            with torch.no_grad():
                for n_, b in noise_bufs.items():
                    b[:] = init_noise['backbone.synthesis.' + n_]
                    b.requires_grad = True                                      # w_projector.py:126-128
                for n_, b in noise_bufs2.items():
                    b[:] = init_noise['superresolution.' + n_]
                    b.requires_grad = True                                      # :129-131
            w_opt = w0.clone().requires_grad_(True)
            # (a non-zero start: at exactly zero the gradient along the viewing axis vanishes and Adam's first step follows rounding noise)
            translation_opt = torch.tensor([IO.PIN_TRANSLATION_START], requires_grad=True)
            t255 = (((target + 1) / 2) * 255).unsqueeze(0)
            if t255.shape[2] > 256:
                t255 = torch.nn.functional.interpolate(t255, size=(256, 256), mode='area')
            vgg16 = lambda img, resize_images=False, return_lpips=True: IO.stub_features(img, fw)      # noqa: E731
            init_ext = torch.Tensor([1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 2.7, 0, 0, 0, 1]).reshape(-1, 4, 4)
            intrinsic = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).unsqueeze(0)
            cam_lr = dict(quat=hp.cam_lr_quat, euler=hp.cam_lr_2d)
            cam_lr['6d'] = hp.cam_lr_6d
            ns = dict(torch=torch, F=torch.nn.functional, np=np, math=math, os=os, PIL=None, tqdm=lambda x: x, global_config=gc, hyperparameters=hp,
                      compute_rotation_matrix_from_quaternion=ref_cam.compute_rotation_matrix_from_quaternion, rot6d_to_rotmat=ref_cam.rot6d_to_rotmat,
                      euler2rot=ref_cam.euler2rot, calc_warping_loss=calc_warping_loss, ray_generator=RaySampler(), G=Gs, vgg16=vgg16,
                      torch_vgg=_stub_torch_vgg(fw), layers='14', cam_predictor=cam_predictor, target_images=t255, target_images_contiguous=target.contiguous(),
                      target_features=vgg16(t255), init_ext=init_ext, intrinsic=intrinsic, canonical_cam=torch.cat([init_ext.reshape(-1, 16), intrinsic], -1),
                      radius=2.7, w_opt=w_opt, translation_opt=translation_opt, noise_bufs=noise_bufs, noise_bufs2=noise_bufs2,
                      optimizer=torch.optim.Adam([w_opt] + list(noise_bufs.values()) + list(noise_bufs2.values()), betas=(0.9, 0.999), lr=hp.first_inv_lr),
                      cam_optimizer=torch.optim.Adam(cam_predictor.parameters(), lr=cam_lr[mode], betas=(0.9, 0.999)),
                      translation_optimizer=torch.optim.Adam([translation_opt], lr=hp.translation_lr),
                      num_steps=PROJ_STEPS, w_std=IO.PIN_W_STD, initial_learning_rate=0.01, lr_rampdown_length=0.25, initial_noise_factor=0.05,
                      noise_ramp_length=0.75, lr_rampup_length=0.05, regularize_noise_weight=1e5, outdir=None, w_name='pin', _trace=[], _gtrace=[],
                      _psnr=lambda im: float(O.psnr_01(im.detach(), target[None])))
            q = list(wns[PROJ_PREHEAT:])
            torch.randn_like = lambda x, **k: q.pop(0).reshape(x.shape)          # w_noise = torch.randn_like(w_opt) (:183)
            try:
                _exec_nodes([node], ns, 'w_projector.project loop')
            finally:
                torch.randn_like = o_randn_like
            trace = torch.tensor(ns['_trace'])
            # ---- the oracle must reproduce it ------------------------------------------------------------------------------------
            po = IO.ProjectorOracle(P, cfg, target[None], num_steps=PROJ_STEPS, optimize_pose=True, use_warping_loss=True, init_noise=init_noise,
                                    w_start=w0, cam_preheat_steps=PROJ_PREHEAT, pose_mode=mode, pose_net=IO.StubPoseNet(base, seed=7), w_std=IO.PIN_W_STD, translation_start=IO.PIN_TRANSLATION_START,
                                    cam_lr=cam_lr[mode])
            otrace = []
            for k in range(PROJ_STEPS):
                r = po.step(*uniforms[k], w_noise=wns[k])
                otrace.append((float(r['loss']), float(r['dist']), float(r['reg']), float(r['warp']), float(O.psnr_01(r['image'], target[None]))))
            otrace = torch.tensor(otrace)
            e = [check(otrace[:, j], trace[:, j], 2e-5, f'projector loop {mode}: {nm}') for j, nm in enumerate(('loss', 'dist', 'reg', 'warp', 'psnr'))]
            check(po.w_opt, w_opt, 1e-5, f'projector loop {mode}: w_opt')
            check(po.translation_opt, translation_opt, 1e-5, f'projector loop {mode}: translation')
            check(po.pose_net.base, cam_predictor.base, 1e-6, f'projector loop {mode}: pose base')
            check(po.pose_net.A, cam_predictor.A, 1e-6, f'projector loop {mode}: pose A')
            dpose = (cam_predictor.base.detach() - base).abs().max().item()
            assert dpose > 0 and (translation_opt.detach().abs().max().item() > 0), 'pose chain received no gradient'
            for k_, b in noise_bufs.items():
                check_adam(po.P['backbone.synthesis.' + k_], b, 1e-5, PROJ_STEPS * 0.01, f'projector loop {mode}: {k_}')
            for k_, b in noise_bufs2.items():
                check_adam(po.P['superresolution.' + k_], b, 1e-5, PROJ_STEPS * 0.01, f'projector loop {mode}: SR {k_}')
            print(f'    {mode}: trace errs {["%.1e" % x for x in e]}, |d pose| {dpose:.2e}, final loss {trace[-1, 0]:.4f}')
            out.update({f'{mode}_trace': trace, f'{mode}_w_opt': w_opt, f'{mode}_translation': translation_opt, f'{mode}_pose_base0': base,
                        f'{mode}_pose_base': cam_predictor.base, f'{mode}_pose_A': cam_predictor.A,
                        f'{mode}_buf_last': list(noise_bufs.values())[-1], f'{mode}_srbuf_last': list(noise_bufs2.values())[-1]})
    finally:
        gc.use_quaternions, gc.use_6d, gc.visualize_opt_process, gc.visualize_warp_process, hp.cam_preheat_steps = saved
    out['w0'] = w0
    save('projector_loop', 2e-5, **out)


C3_FULL_STEPS, C3_FULL_PREHEAT = 2, 1


def gen_c3_full():
    """Config C3 at FULL size (BASELINE.json configs[2]): one camera-preheat step + one full step of the reference's optimisation-loop body
    (w_projector.py:145-270, executed as is) on the ffhqrebalanced512-128-shaped generator built from the reference's own classes --
    quaternion pose chain, 512^2 synthesis, the reference's calc_warping_loss (second, canonical-view forward; depth re-projection), LPIPS-
    stub distance, noise regulariser, the three Adam optimisers.  Recorded: the loss terms of both steps, the gradients that reach the
    translation, the pose parameters and (probes of) the latent, and where the optimisers put them."""
    print('config C3 at full size (reference loop body, lifted) -- takes a few minutes')
    from utils import camera_utils as ref_cam
    from training.warping_loss import calc_warping_loss
    from configs import global_config as gc, hyperparameters as hp
    from oracle import inversion_oracle as IO
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)
    fw = IO.stub_feature_weights()
    loop = [n for n in _func(_parse('training/projectors/w_projector.py'), 'project').body if isinstance(n, ast.For)][-1]
    node = _lifted_loop(loop, 'num_steps', "_trace.append((float(loss), float(dist), float(reg_loss), float(warp_loss), _psnr(pred_dict['image']))); "
                        "_gtrace.append((translation_opt.grad.clone(), cam_predictor.base.grad.clone(), None if w_opt.grad is None else w_opt.grad.clone()))")
    saved = (gc.use_quaternions, gc.use_6d, gc.visualize_opt_process, gc.visualize_warp_process, hp.cam_preheat_steps)
    o_randn_like = torch.randn_like
    mode = 'quat'
    try:
        gc.visualize_opt_process = gc.visualize_warp_process = False
        hp.cam_preheat_steps = C3_FULL_PREHEAT
        gc.use_quaternions, gc.use_6d = True, False
        G = RefComposite(cfg, P).requires_grad_(False)
        pin = IO.pin_projector_inputs(cfg, P, mode)
        uniforms, wns, init_noise, w0, base = pin['uniforms'], pin['wns'], pin['init_noise'], pin['w0'], pin['pose_base']
        cam_predictor = IO.StubPoseNet(base, seed=7)
        Gs = _GSteps(G, uniforms, calls_per_step=2)
        noise_bufs = {n: b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n}
        noise_bufs2 = {f'{blk}.{n}': b for blk in ('block0', 'block1') for n, b in getattr(G, blk).named_buffers() if 'noise_const' in n}
        with torch.no_grad():
            for n_, b in noise_bufs.items():
                b[:] = init_noise['backbone.synthesis.' + n_]
                b.requires_grad = True
            for n_, b in noise_bufs2.items():
                b[:] = init_noise['superresolution.' + n_]
                b.requires_grad = True
        w_opt = w0.clone().requires_grad_(True)
        translation_opt = torch.tensor([IO.PIN_TRANSLATION_START], requires_grad=True)
        t255 = (((target + 1) / 2) * 255).unsqueeze(0)
        if t255.shape[2] > 256:
            t255 = torch.nn.functional.interpolate(t255, size=(256, 256), mode='area')
        vgg16 = lambda img, resize_images=False, return_lpips=True: IO.stub_features(img, fw)      # noqa: E731
        init_ext = torch.Tensor([1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 2.7, 0, 0, 0, 1]).reshape(-1, 4, 4)
        intrinsic = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).unsqueeze(0)
        ns = dict(torch=torch, F=torch.nn.functional, np=np, math=math, os=os, PIL=None, tqdm=lambda x: x, global_config=gc, hyperparameters=hp,
                  compute_rotation_matrix_from_quaternion=ref_cam.compute_rotation_matrix_from_quaternion, rot6d_to_rotmat=ref_cam.rot6d_to_rotmat,
                  euler2rot=ref_cam.euler2rot, calc_warping_loss=calc_warping_loss, ray_generator=RaySampler(), G=Gs, vgg16=vgg16,
                  torch_vgg=_stub_torch_vgg(fw), layers='14', cam_predictor=cam_predictor, target_images=t255, target_images_contiguous=target.contiguous(),
                  target_features=vgg16(t255), init_ext=init_ext, intrinsic=intrinsic, canonical_cam=torch.cat([init_ext.reshape(-1, 16), intrinsic], -1),
                  radius=2.7, w_opt=w_opt, translation_opt=translation_opt, noise_bufs=noise_bufs, noise_bufs2=noise_bufs2,
                  optimizer=torch.optim.Adam([w_opt] + list(noise_bufs.values()) + list(noise_bufs2.values()), betas=(0.9, 0.999), lr=hp.first_inv_lr),
                  cam_optimizer=torch.optim.Adam(cam_predictor.parameters(), lr=hp.cam_lr_quat, betas=(0.9, 0.999)),
                  translation_optimizer=torch.optim.Adam([translation_opt], lr=hp.translation_lr),
                  num_steps=C3_FULL_STEPS, w_std=IO.PIN_W_STD, initial_learning_rate=0.01, lr_rampdown_length=0.25, initial_noise_factor=0.05,
                  noise_ramp_length=0.75, lr_rampup_length=0.05, regularize_noise_weight=1e5, outdir=None, w_name='pin', _trace=[], _gtrace=[],
                  _psnr=lambda im: float(O.psnr_01(im.detach(), target[None])))
        q = list(wns[C3_FULL_PREHEAT:])
        torch.randn_like = lambda x, **k: q.pop(0).reshape(x.shape)
        try:
            _exec_nodes([node], ns, 'w_projector.project loop, full size')
        finally:
            torch.randn_like = o_randn_like
        trace = torch.tensor(ns['_trace'])
        gtr = ns['_gtrace']
        print('    reference trace (loss, dist, reg, warp, psnr):', trace.tolist())
        # ---- the oracle must reproduce it ----------------------------------------------------------------------------------------
        po = IO.ProjectorOracle(P, cfg, target[None], num_steps=C3_FULL_STEPS, optimize_pose=True, use_warping_loss=True, init_noise=init_noise,
                                w_start=w0, cam_preheat_steps=C3_FULL_PREHEAT, pose_mode=mode, pose_net=IO.StubPoseNet(base, seed=7), w_std=IO.PIN_W_STD,
                                translation_start=IO.PIN_TRANSLATION_START, cam_lr=hp.cam_lr_quat)
        otrace = []
        for k in range(C3_FULL_STEPS):
            r = po.step(*uniforms[k], w_noise=wns[k])
            otrace.append((float(r['loss']), float(r['dist']), float(r['reg']), float(r['warp']), float(O.psnr_01(r['image'], target[None]))))
        otrace = torch.tensor(otrace)
        e = [check(otrace[:, j], trace[:, j], 5e-5, f'C3 full: {nm}') for j, nm in enumerate(('loss', 'dist', 'reg', 'warp', 'psnr'))]
        check(po.w_opt, w_opt, 1e-5, 'C3 full: w_opt')
        check(po.translation_opt, translation_opt, 1e-5, 'C3 full: translation')
        check(po.pose_net.base, cam_predictor.base, 1e-6, 'C3 full: pose base')
        print(f'    trace errs {["%.1e" % x for x in e]}')
        dw = gtr[-1][2].flatten()
        probe = torch.Generator().manual_seed(77)
        idx = torch.randint(0, dw.numel(), (256,), generator=probe)
        save('c3_full', 5e-5, trace=trace, d_translation=torch.stack([g[0] for g in gtr]), d_pose_base=torch.stack([g[1] for g in gtr]),
             dw_idx=idx, dw_val=dw[idx], dw_stat=np.array([dw.norm().item(), dw.abs().max().item()]), w_opt=w_opt, translation=translation_opt,
             pose_base0=base, pose_base=cam_predictor.base, pose_A=cam_predictor.A, target_probe=target.flatten()[::4099].clone())
    finally:
        gc.use_quaternions, gc.use_6d, gc.visualize_opt_process, gc.visualize_warp_process, hp.cam_preheat_steps = saved


C2_FULL_STEPS = 10


def gen_c2_full():
    """Config C2 at FULL size over a trajectory (BASELINE.json configs[1]): ten steps of the reference's optimisation-loop body
    (w_projector.py:145-270, executed as is) on the ffhqrebalanced512-128-shaped generator built from the reference's own classes, with the
    camera held fixed the way config C2 means it -- the loop body always runs the pose chain and the warping loss, so here calc_warping_loss
    is a stub returning (None, None) (the body's `if warp_loss != None` then skips the term) and the two pose optimisers run with lr = 0 (Adam
    moves nothing), no camera preheat.  Recorded: loss / distance / regulariser / PSNR of every step, the camera the chain produces, the final
    latent and probes of two noise maps."""
    print('config C2 at full size, %d steps (reference loop body, lifted) -- takes several minutes' % C2_FULL_STEPS)
    from utils import camera_utils as ref_cam
    from configs import global_config as gc, hyperparameters as hp
    from oracle import inversion_oracle as IO
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)
    fw = IO.stub_feature_weights()
    loop = [n for n in _func(_parse('training/projectors/w_projector.py'), 'project').body if isinstance(n, ast.For)][-1]
    node = _lifted_loop(loop, 'num_steps', "_trace.append((float(loss), float(dist), float(reg_loss), _psnr(pred_dict['image']))); _cams.append(pred_cam.detach().clone())")
    saved = (gc.use_quaternions, gc.use_6d, gc.visualize_opt_process, gc.visualize_warp_process, hp.cam_preheat_steps)
    o_randn_like = torch.randn_like
    mode = 'quat'
    try:
        gc.visualize_opt_process = gc.visualize_warp_process = False
        hp.cam_preheat_steps = 0
        gc.use_quaternions, gc.use_6d = True, False
        G = RefComposite(cfg, P).requires_grad_(False)
        pin = IO.pin_projector_inputs(cfg, P, mode, steps=C2_FULL_STEPS)
        uniforms, wns, init_noise, w0, base = pin['uniforms'], pin['wns'], pin['init_noise'], pin['w0'], pin['pose_base']
        cam_predictor = IO.StubPoseNet(base, seed=7)
        Gs = _GSteps(G, uniforms, calls_per_step=1)
        noise_bufs = {n: b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n}
        noise_bufs2 = {f'{blk}.{n}': b for blk in ('block0', 'block1') for n, b in getattr(G, blk).named_buffers() if 'noise_const' in n}
        with torch.no_grad():
            for n_, b in noise_bufs.items():
                b[:] = init_noise['backbone.synthesis.' + n_]
                b.requires_grad = True
            for n_, b in noise_bufs2.items():
                b[:] = init_noise['superresolution.' + n_]
                b.requires_grad = True
        w_opt = w0.clone().requires_grad_(True)
        translation_opt = torch.tensor([IO.PIN_TRANSLATION_START], requires_grad=True)
        t255 = (((target + 1) / 2) * 255).unsqueeze(0)
        if t255.shape[2] > 256:
            t255 = torch.nn.functional.interpolate(t255, size=(256, 256), mode='area')
        vgg16 = lambda img, resize_images=False, return_lpips=True: IO.stub_features(img, fw)      # noqa: E731
        init_ext = torch.Tensor([1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 2.7, 0, 0, 0, 1]).reshape(-1, 4, 4)
        intrinsic = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).unsqueeze(0)
        ns = dict(torch=torch, F=torch.nn.functional, np=np, math=math, os=os, PIL=None, tqdm=lambda x: x, global_config=gc, hyperparameters=hp,
                  compute_rotation_matrix_from_quaternion=ref_cam.compute_rotation_matrix_from_quaternion, rot6d_to_rotmat=ref_cam.rot6d_to_rotmat,
                  euler2rot=ref_cam.euler2rot, calc_warping_loss=lambda *a, **k: (None, None), ray_generator=RaySampler(), G=Gs, vgg16=vgg16,
                  torch_vgg=None, layers='14', cam_predictor=cam_predictor, target_images=t255, target_images_contiguous=target.contiguous(),
                  target_features=vgg16(t255), init_ext=init_ext, intrinsic=intrinsic, canonical_cam=torch.cat([init_ext.reshape(-1, 16), intrinsic], -1),
                  radius=2.7, w_opt=w_opt, translation_opt=translation_opt, noise_bufs=noise_bufs, noise_bufs2=noise_bufs2,
                  optimizer=torch.optim.Adam([w_opt] + list(noise_bufs.values()) + list(noise_bufs2.values()), betas=(0.9, 0.999), lr=hp.first_inv_lr),
                  cam_optimizer=torch.optim.Adam(cam_predictor.parameters(), lr=0.0, betas=(0.9, 0.999)),
                  translation_optimizer=torch.optim.Adam([translation_opt], lr=0.0),
                  num_steps=C2_FULL_STEPS, w_std=IO.PIN_W_STD, initial_learning_rate=0.01, lr_rampdown_length=0.25, initial_noise_factor=0.05,
                  noise_ramp_length=0.75, lr_rampup_length=0.05, regularize_noise_weight=1e5, outdir=None, w_name='pin', _trace=[], _cams=[],
                  _psnr=lambda im: float(O.psnr_01(im.detach(), target[None])))
        q = list(wns)
        torch.randn_like = lambda x, **k: q.pop(0).reshape(x.shape)
        try:
            _exec_nodes([node], ns, 'w_projector.project loop, full size, C2')
        finally:
            torch.randn_like = o_randn_like
        trace = torch.tensor(ns['_trace'])
        cams = torch.cat(ns['_cams'])
        assert float((cams - cams[:1]).abs().max()) == 0.0, 'the camera moved: lr = 0 must hold it'
        cam = cams[:1]
        print('    reference trace (loss, dist, reg, psnr):', trace.tolist())
        # ---- the oracle must reproduce it ----------------------------------------------------------------------------------------
        po = IO.ProjectorOracle(P, cfg, target[None], num_steps=C2_FULL_STEPS, cam=cam, init_noise=init_noise, w_start=w0, w_std=IO.PIN_W_STD)
        otrace = []
        for k in range(C2_FULL_STEPS):
            r = po.step(*uniforms[k], w_noise=wns[k])
            otrace.append((float(r['loss']), float(r['dist']), float(r['reg']), float(O.psnr_01(r['image'], target[None]))))
        otrace = torch.tensor(otrace)
        e = [check(otrace[:, j], trace[:, j], 5e-5, f'C2 full: {nm}') for j, nm in enumerate(('loss', 'dist', 'reg', 'psnr'))]
        check(po.w_opt, w_opt, 1e-5, 'C2 full: w_opt')
        print(f'    trace errs {["%.1e" % x for x in e]}')
        probe = torch.Generator().manual_seed(78)
        nb_last, sb_last = list(noise_bufs.values())[-1].detach().flatten(), list(noise_bufs2.values())[-1].detach().flatten()
        i1, i2 = torch.randint(0, nb_last.numel(), (256,), generator=probe), torch.randint(0, sb_last.numel(), (256,), generator=probe)
        save('c2_full', 5e-5, trace=trace, cam=cam, w_opt=w_opt, buf_idx=i1, buf_val=nb_last[i1], srbuf_idx=i2, srbuf_val=sb_last[i2],
             target_probe=target.flatten()[::4099].clone())
    finally:
        gc.use_quaternions, gc.use_6d, gc.visualize_opt_process, gc.visualize_warp_process, hp.cam_preheat_steps = saved


C4_FULL_STEPS = 5


def gen_c4_full():
    """Config C4 at FULL size over a trajectory (BASELINE.json configs[3]): five steps of SingleIDCoach.train's inner loop
    (single_id_coach.py:64-77) with BaseCoach.calc_loss / forward (base_coach.py:101-126,162-164), lifted and executed as is on the
    ffhqrebalanced512-128-shaped generator of the reference's classes (Adam 3e-4 over all 30.7 M weights, noise_mode='random' replayed)."""
    print('config C4 at full size, %d steps (reference loop body, lifted) -- takes several minutes' % C4_FULL_STEPS)
    from configs import global_config as gc, hyperparameters as hp
    from criteria import l2_loss
    from oracle import inversion_oracle as IO
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)[None]
    fw = IO.stub_feature_weights()
    base = _parse('training/coaches/base_coach.py')
    ns_c = dict(torch=torch, F=torch.nn.functional, hyperparameters=hp, global_config=gc, l2_loss=l2_loss, wandb=None)
    _exec_nodes([_func(base, 'compute_tv_norm'), _func(base, 'calc_loss', 'BaseCoach'), _func(base, 'forward', 'BaseCoach')], ns_c, 'BaseCoach')
    train = _func(_parse('training/coaches/single_id_coach.py'), 'train', 'SingleIDCoach')
    loops = [n for n in ast.walk(train) if isinstance(n, ast.For) and 'max_pti_steps' in ast.unparse(n.iter)]
    assert len(loops) == 1
    node = _lifted_loop(loops[0], 'hyperparameters.max_pti_steps', "_trace.append((float(loss), float(l2_loss_val), float(loss_lpips), _psnr(generated_images['image'])))")
    pin = IO.pin_tuner_inputs(cfg, steps=C4_FULL_STEPS)
    w_pivot, cam, noise_names, uniforms, noises = pin['w_pivot'], pin['cam'], pin['noise_names'], pin['uniforms'], pin['noises']
    out = dict(target_probe=target.flatten()[::4099].clone(), w_pivot=w_pivot, cam=cam)
    saved = (hp.max_pti_steps, hp.LPIPS_value_threshold, gc.training_step)
    try:
        hp.max_pti_steps, hp.LPIPS_value_threshold = C4_FULL_STEPS, -1.0
        G = RefComposite(cfg, P).requires_grad_(True)
        Gs = _GSteps(G, uniforms, calls_per_step=1, randn=[[nz[nm] for nm in noise_names] for nz in noises])
        coach = types.SimpleNamespace(G=Gs, use_wandb=False, space_regulizer=None,
                                      lpips_loss=lambda a, b: (IO.stub_features(a, fw) - IO.stub_features(b, fw)).square().sum())
        coach.calc_loss = types.MethodType(ns_c['calc_loss'], coach)
        coach.forward = types.MethodType(ns_c['forward'], coach)
        coach.optimizer = torch.optim.Adam(G.parameters(), lr=hp.pti_learning_rate)
        ns = dict(self=coach, tqdm=lambda x: x, hyperparameters=hp, global_config=gc, w_pivot=w_pivot, freezed_cam=cam, real_images_batch=target,
                  image_name='pin', use_ball_holder=True, log_images_counter=0, _trace=[], _psnr=lambda im: float(O.psnr_01(im.detach(), target)))
        _exec_nodes([node], ns, 'SingleIDCoach.train loop, full size')
        trace = torch.tensor(ns['_trace'])
        print('    reference trace (loss, l2, lpips, psnr):', trace.tolist())
        to = IO.PivotalTunerOracle(P, cfg, target, w_pivot, cam, lr=hp.pti_learning_rate, lpips_threshold=-1.0)
        otrace = []
        for k in range(C4_FULL_STEPS):
            r = to.step(*uniforms[k], noise_mode='random', noises=noises[k], early_stop=True)
            otrace.append((float(r['loss']), float(r['l2']), float(r['lpips']), float(O.psnr_01(r['image'], target))))
        otrace = torch.tensor(otrace)
        e = [check(otrace[:, j], trace[:, j], 5e-5, f'C4 full: {nm}') for j, nm in enumerate(('loss', 'l2', 'lpips', 'psnr'))]
        sd = dict(G.named_parameters())
        probe = torch.Generator().manual_seed(79)
        for k_ in TUNER_KEYS:
            rk = k_[len('superresolution.'):] if k_.startswith('superresolution.') else k_
            check(to.P[k_], sd[rk], 1e-5, f'C4 full: {k_}')
            flat, flat0 = sd[rk].detach().flatten(), P[k_].flatten()
            idx = torch.randint(0, flat.numel(), (min(256, flat.numel()),), generator=probe)
            out[f'p_idx.{k_}'], out[f'p_val.{k_}'], out[f'p_move.{k_}'] = idx, flat[idx], np.float64((flat - flat0).abs().max().item())
        print(f'    {trace.shape[0]} updates, trace errs {["%.1e" % x for x in e]}, loss {trace[0, 0]:.4f} -> {trace[-1, 0]:.4f}')
        out['trace'] = trace
    finally:
        hp.max_pti_steps, hp.LPIPS_value_threshold, gc.training_step = saved
    save('c4_full', 5e-5, **out)


TUNER_KEYS = ('backbone.synthesis.b8.conv0.weight', 'backbone.synthesis.b16.torgb.bias', 'backbone.synthesis.b32.conv1.noise_strength',
              'backbone.synthesis.b16.conv1.affine.weight', 'superresolution.block1.conv1.weight', 'superresolution.block0.torgb.weight',
              'decoder.net.0.weight', 'decoder.net.2.bias')


def gen_tuner_loop():
    """Phase B: SingleIDCoach.train's inner loop (single_id_coach.py:64-77) with BaseCoach.calc_loss / forward (base_coach.py:101-126,
    162-164) and compute_tv_norm (:294-305), all lifted and executed as is; PivotalTunerOracle must reproduce losses and weights."""
    print('pivotal-tuning loop (reference loop body, lifted)')
    from configs import global_config as gc, hyperparameters as hp
    from criteria import l2_loss
    from oracle import inversion_oracle as IO
    cfg = IO.pin_config(tuner=True)                        # calc_loss hard-codes the 128^2 raw image (base_coach.py:103)
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)[None]
    TUNER_STEPS = IO.PIN_TUNER_STEPS
    fw = IO.stub_feature_weights()
    base = _parse('training/coaches/base_coach.py')
    ns_c = dict(torch=torch, F=torch.nn.functional, hyperparameters=hp, global_config=gc, l2_loss=l2_loss, wandb=None)
    _exec_nodes([_func(base, 'compute_tv_norm'), _func(base, 'calc_loss', 'BaseCoach'), _func(base, 'forward', 'BaseCoach')], ns_c, 'BaseCoach')
    train = _func(_parse('training/coaches/single_id_coach.py'), 'train', 'SingleIDCoach')
    loops = [n for n in ast.walk(train) if isinstance(n, ast.For) and 'max_pti_steps' in ast.unparse(n.iter)]
    assert len(loops) == 1
    node = _lifted_loop(loops[0], 'hyperparameters.max_pti_steps', "_trace.append((float(loss), float(l2_loss_val), float(loss_lpips), _psnr(generated_images['image'])))")
    pin = IO.pin_tuner_inputs(cfg)
    w_pivot, cam, noise_names, uniforms, noises = pin['w_pivot'], pin['cam'], pin['noise_names'], pin['uniforms'], pin['noises']
    out = dict(target_probe=target.flatten()[::997].clone(), w_pivot=w_pivot, cam=cam)
    saved = (hp.max_pti_steps, hp.LPIPS_value_threshold, gc.training_step)
    try:
        for tag, thr in (('full', -1.0), ('stop', None)):
            if thr is None:         # a threshold that trips in the middle of the run: between the recorded LPIPS values of steps 2 and 3
                lp = out['full_trace'][:, 2]
                assert lp[3] < lp[2], lp
                thr = float((lp[2] + lp[3]) / 2)
            hp.max_pti_steps, hp.LPIPS_value_threshold = TUNER_STEPS, thr
            G = RefComposite(cfg, P).requires_grad_(True)
            Gs = _GSteps(G, uniforms, calls_per_step=1, randn=[[nz[nm] for nm in noise_names] for nz in noises])
            coach = types.SimpleNamespace(G=Gs, use_wandb=False, space_regulizer=None,
                                          lpips_loss=lambda a, b: (IO.stub_features(a, fw) - IO.stub_features(b, fw)).square().sum())
            coach.calc_loss = types.MethodType(ns_c['calc_loss'], coach)
            coach.forward = types.MethodType(ns_c['forward'], coach)
            coach.optimizer = torch.optim.Adam(G.parameters(), lr=hp.pti_learning_rate)               # base_coach.py:96-99
            ns = dict(self=coach, tqdm=lambda x: x, hyperparameters=hp, global_config=gc, w_pivot=w_pivot, freezed_cam=cam, real_images_batch=target,
                      image_name='pin', use_ball_holder=True, log_images_counter=0, _trace=[],
                      _psnr=lambda im: float(O.psnr_01(im.detach(), target)))
            _exec_nodes([node], ns, 'SingleIDCoach.train loop')
            trace = torch.tensor(ns['_trace'])
            # the early exit `break`s before the trace statement: the number of recorded rows is the number of completed updates
            to = IO.PivotalTunerOracle(P, cfg, target, w_pivot, cam, lr=hp.pti_learning_rate, lpips_threshold=thr)
            otrace = []
            for k in range(TUNER_STEPS):
                r = to.step(*uniforms[k], noise_mode='random', noises=noises[k], early_stop=True)
                if r['done']:
                    break
                otrace.append((float(r['loss']), float(r['l2']), float(r['lpips']), float(O.psnr_01(r['image'], target))))
            otrace = torch.tensor(otrace)
            assert otrace.shape == trace.shape, (tag, otrace.shape, trace.shape)
            e = [check(otrace[:, j], trace[:, j], 2e-5, f'tuner loop {tag}: {nm}') for j, nm in enumerate(('loss', 'l2', 'lpips', 'psnr'))]
            sd = dict(G.named_parameters())
            for k_ in TUNER_KEYS:
                rk = k_[len('superresolution.'):] if k_.startswith('superresolution.') else k_
                check(to.P[k_], sd[rk], 1e-5, f'tuner loop {tag}: {k_}')
                out[f'{tag}_p.{k_}'] = sd[rk]
            print(f'    {tag}: {trace.shape[0]} updates, trace errs {["%.1e" % x for x in e]}, loss {trace[0, 0]:.4f} -> {trace[-1, 0]:.4f}')
            out[f'{tag}_trace'] = trace
            out[f'{tag}_thr'] = np.float64(thr)
    finally:
        hp.max_pti_steps, hp.LPIPS_value_threshold, gc.training_step = saved
    save('tuner_loop', 2e-5, **out)


def gen_inference():
    """Inference consumers: orbit cameras, grid samples, density grid, mean-latent statistics."""
    print('inference consumers')
    import ast
    from utils import camera_utils as ref_cam
    from oracle import inference_oracle as IO
    out = {}
    # LookAtPoseSampler (utils/camera_utils.py:87-105)
    hv = [(3.14 / 2, 3.14 / 2), (3.14 / 2 + 0.35, 3.14 / 2 - 0.05), (1.0, 2.0), (2.2, 0.7)]
    poses = torch.cat([ref_cam.LookAtPoseSampler.sample(h, v, torch.tensor([0., 0, 0]), radius=2.7) for h, v in hv])
    for (h, v), m in zip(hv, poses):
        check(IO.lookat_pose(h, v, (0., 0., 0.), 2.7), m, 1e-6, 'lookat pose')
    out.update(dict(hv=torch.tensor(hv), poses=poses))
    # orbit of gen_interp_video (gen_videos.py:105-117; the file itself needs imageio, so the loop is restated around the reference sampler)
    F_ = 8
    K = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]])
    cams = []
    for frame_idx in range(F_):
        m = ref_cam.LookAtPoseSampler.sample(3.14 / 2 + 0.35 * np.sin(2 * 3.14 * frame_idx / F_), 3.14 / 2 - 0.05 + 0.25 * np.cos(2 * 3.14 * frame_idx / F_),
                                             torch.tensor([0., 0, 0]), radius=2.7)
        cams.append(torch.cat([m.reshape(-1, 16), K.reshape(-1, 9)], 1))
    cams = torch.cat(cams)
    check(IO.orbit_cameras(F_), cams, 1e-6, 'orbit cameras')
    out['orbit8'] = cams
    # create_samples: the function's own source, lifted out of single_id_coach.py (the module imports wandb/lpips/mrcfile)
    src = open(os.path.join(REF, 'training/coaches/single_id_coach.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'create_samples'][0]
    ns = dict(np=np, torch=torch)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'create_samples', 'exec'), ns)
    for N, L_ in ((6, 1.0), (20, 1.0), (37, 2.0)):
        ref_s, _, _ = ns['create_samples'](N=N, voxel_origin=[0, 0, 0], cube_length=L_)
        check(IO.create_samples(N, L_), ref_s, 1e-6, f'create_samples {N}')
    out['samples20'] = ns['create_samples'](N=20, voxel_origin=[0, 0, 0], cube_length=1.0)[0]
    # density grid: create_geometry's evaluation (single_id_coach.py:120-157) on the small reference generator
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=0)
    G = RefComposite(cfg, P)
    ws = O.synth_ws(cfg, 1, seed=1, wplus=True)
    res = 12
    with torch.no_grad():
        samples, _, _ = ns['create_samples'](N=res, voxel_origin=[0, 0, 0], cube_length=cfg.rendering['box_warp'] * 1)
        planes = G.backbone.synthesis(ws, update_emas=False, noise_mode='const', force_fp32=True)
        planes = planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])
        dirs = torch.zeros((1, samples.shape[1], 3)); dirs[..., -1] = -1
        sigma = G.renderer.run_model(planes, G.decoder, samples, dirs, cfg.rendering)['sigma']
    sigmas = np.flip(sigma.reshape((res, res, res)).numpy(), 0).copy()
    pad = int(30 * res / 256)
    for sl in ((slice(None, pad),), (slice(-pad, None),), (slice(None), slice(None, pad)), (slice(None), slice(-pad, None)),
               (slice(None), slice(None), slice(None, pad)), (slice(None), slice(None), slice(-pad, None))):
        sigmas[sl] = -1000
    check(IO.density_grid(P, cfg, ws, res), sigmas, 2e-5, 'density grid')
    out.update(dict(grid_ws=ws, grid12=sigmas))
    # mean latent (w_projector.py:88-97) on the small mapping network
    n = 64
    ext = ref_cam.euler2rot(torch.tensor([math.pi / 2]), torch.tensor([math.pi / 2]), torch.zeros(1, 1), batch_size=1)
    cam_init = torch.cat([ext.reshape(1, 16), torch.tensor([[4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]])], -1)
    z = np.random.RandomState(123).randn(n, cfg.z_dim)
    with torch.no_grad():
        w = G.backbone.mapping(torch.from_numpy(z).float(), cam_init.repeat(n, 1), truncation_cutoff=14, truncation_psi=0.7)
    w = w[:, :1, :].numpy().astype(np.float32)
    w_avg = np.mean(w, axis=0, keepdims=True)
    w_std = (np.sum((w - w_avg) ** 2) / n) ** 0.5
    oa, os_ = IO.w_stats(P, cfg, n)
    check(oa, w_avg, 1e-5, 'w_avg')
    assert abs(os_ - w_std) <= 1e-5 * max(1.0, w_std), (os_, w_std)
    out.update(dict(w_avg64=w_avg, w_std64=np.float32(w_std)))
    save('inference', 2e-5, **out)


GRAD_KEYS_POSE = ('conv1.weight', 'bn1.weight', 'bn1.bias', 'layer1.0.conv2.weight', 'layer2.0.downsample.0.weight', 'layer2.0.downsample.1.weight',
                  'layer3.2.bn1.bias', 'layer4.2.conv1.weight', 'fc.weight', 'fc3.bias')


def gen_pose_net():
    """In-loop pose estimator: the reference's own ResNet class (scripts/resnet/resnet.py) in eval mode."""
    print('pose estimator (ResNet-34)')
    sys.path.insert(0, os.path.join(REF, 'scripts'))
    from resnet import resnet as ref_resnet
    from oracle import pose_net_oracle as PO
    out = {}
    for dims in (4, 6):
        net = ref_resnet.resnet34(output_dims=dims).eval()
        sd = PO.synth_state(seed=3, output_dims=dims)
        net.load_state_dict(sd, strict=True)
        img = O._randn('pose_img', dims, (2, 3, 64, 64)).clamp(-1, 1)
        y = net(img)
        gy = O._randn('pose_gy', dims, y.shape)
        params = dict(net.named_parameters())
        grads = torch.autograd.grad(y, [params[k] for k in GRAD_KEYS_POSE], gy)
        sdo = {k: (v.clone().requires_grad_(True) if k in GRAD_KEYS_POSE else v) for k, v in sd.items()}
        yo = PO.forward(sdo, img)
        check(yo, y, 1e-6, f'pose net output d={dims}')
        go = torch.autograd.grad(yo, [sdo[k] for k in GRAD_KEYS_POSE], gy)
        for k, a, b in zip(GRAD_KEYS_POSE, go, grads):
            check(a, b, 1e-5, f'pose net grad {k}')
        out.update({f'd{dims}_img': img, f'd{dims}_y': y, f'd{dims}_gy': gy})
        for k, g in zip(GRAD_KEYS_POSE, grads):
            if g.numel() <= 40000:
                out[f'd{dims}_g.{k}'] = g
            else:                           # large tensors: a strided sample
                out[f'd{dims}_gs.{k}'] = g.flatten()[::97].clone()
    save('pose_net', 1e-5, **out)


def gen_e4e():
    """One-shot latent encoder: the reference's own Encoder4Editing(50, 'ir_se') in eval mode.  Its module
    imports models.e4e.stylegan2.model, whose `op` sub-package JIT-compiles CUDA extensions at import; the encoder uses none of them
    (EqualLinear without activation), so the sub-package is replaced by an empty stand-in for the import."""
    print('e4e encoder (IR-SE-50 + 18 style heads)')
    import types as _types
    from oracle import e4e_oracle as EO
    stub = _types.ModuleType('models.e4e.stylegan2.op')
    for nm in ('fused_act', 'upfirdn2d'):
        sub = _types.ModuleType(f'models.e4e.stylegan2.op.{nm}')
        sub.FusedLeakyReLU = sub.fused_leaky_relu = sub.upfirdn2d = None
        sys.modules[f'models.e4e.stylegan2.op.{nm}'] = sub
        setattr(stub, nm, sub)
    stub.FusedLeakyReLU = stub.fused_leaky_relu = stub.upfirdn2d = None
    stub.__path__ = []
    sys.modules['models.e4e.stylegan2.op'] = stub
    from models.e4e.encoders.psp_encoders import Encoder4Editing
    net = Encoder4Editing(50, 'ir_se').eval()
    sd = EO.synth_state(seed=5)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    img = O._randn('e4e_img', 5, (2, 3, 64, 64)).clamp(-1, 1) * 127.5 + 127.5          # the projector feeds the [0,255] image (w_projector.py:71-74,100)
    with torch.no_grad():
        y = net(img)
        yo = EO.forward(sd, img)
    check(yo, y, 2e-5, 'e4e codes')
    assert float(y[:, 0].abs().max()) > 1e-3 and float((y[:, 3] - y[:, 0]).abs().max()) > 1e-3
    save('e4e', 2e-5, img=img, codes=y)


def gen_sr_heads():
    """The super-resolution heads other than 8XDC (training/superresolution.py:29-152): SuperresolutionHybrid8X / 4X / 2X / Deepfp32, each
    instantiated from the reference's own class with the oracle's deterministic weights, forward + gradients on seeded inputs.  The 4X / 2X /
    Deepfp32 heads start with a SynthesisBlockNoUp (:155-262).  Stored: the inputs' seeds are implicit (O._randn keys), probes of the output and
    of every gradient."""
    from training import superresolution as ref_sr
    classes = {'8X': ref_sr.SuperresolutionHybrid8X, '4X': ref_sr.SuperresolutionHybrid4X, '2X': ref_sr.SuperresolutionHybrid2X,
               'Deepfp32': ref_sr.SuperresolutionHybridDeepfp32}
    probe = torch.Generator().manual_seed(123)
    arrays = {}
    for kind, cls in classes.items():
        in_res, widths, up0, out_res, rule, follows = O.SR_HEADS[kind]
        kw = dict(channels=32, img_resolution=out_res, sr_num_fp16_res=4, channel_base=32768, channel_max=512, fused_modconv_default='inference_only')
        if kind != 'Deepfp32':
            kw['sr_antialias'] = True
        head = cls(**kw).eval().float()
        P = O.sr_head_params(kind, seed=3)
        sd = {k[len('superresolution.'):]: v for k, v in P.items()}
        missing, unexpected = head.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (kind, missing, unexpected)
        # two input sizes per head: its own input resolution, and a smaller one that goes through the bilinear resize (antialias where the head passes it)
        for tag, r in (('own', in_res), ('small', in_res // 2)):
            x = O._randn(f'srx.{kind}.{tag}', 5, (1, 32, r, r)).requires_grad_(True)
            rgb = O._randn(f'srrgb.{kind}.{tag}', 5, (1, 3, r, r)).requires_grad_(True)
            ws = O._randn(f'srws.{kind}', 5, (1, 14, 512)).requires_grad_(True)
            leaves = [head.block0.conv0.weight, head.block0.conv1.noise_strength, head.block1.conv0.weight, head.block1.torgb.weight, head.block1.torgb.bias]
            names = ['block0.conv0.weight', 'block0.conv1.noise_strength', 'block1.conv0.weight', 'block1.torgb.weight', 'block1.torgb.bias']
            img = head(rgb * 1.0, x, ws, noise_mode='const', force_fp32=True)       # (the no-up block adds onto its image argument in place: not a leaf)
            assert tuple(img.shape) == (1, 3, out_res, out_res)
            g = O._randn(f'srg.{kind}', 5, img.shape) / img.numel() ** 0.5
            grads = torch.autograd.grad(img, [x, rgb, ws] + leaves, g)
            Pg = {k: v.clone().requires_grad_(k[len('superresolution.'):] in names) for k, v in P.items()}
            xo, ro, wo = x.detach().clone().requires_grad_(True), rgb.detach().clone().requires_grad_(True), ws.detach().clone().requires_grad_(True)
            io = O.sr_head(Pg, kind, ro, xo, wo, sr_antialias=True, conv_clamp=256.0, noise_mode='const')
            go = torch.autograd.grad(io, [xo, ro, wo] + [Pg['superresolution.' + n] for n in names], g)
            e = check(io, img, 1e-5, f'sr head {kind} {tag} image')
            for nm, a, b in zip(['x', 'rgb', 'ws'] + names, go, grads):
                check(a, b, 2e-5, f'sr head {kind} {tag} d {nm}')
            print(f'    {kind:8s} {tag:5s} {r}^2 -> {out_res}^2: image err {e:.2e}')
            idx = torch.randint(0, img.numel(), (2048,), generator=probe)
            arrays[f'{kind}.{tag}.idx'] = idx
            arrays[f'{kind}.{tag}.img'] = img.flatten()[idx]
            arrays[f'{kind}.{tag}.img_stats'] = np.array([img.mean().item(), img.abs().mean().item(), img.min().item(), img.max().item()])
            for nm, gv in zip(['x', 'rgb', 'ws'] + names, grads):
                flat = gv.detach().flatten()
                gi = torch.randint(0, flat.numel(), (min(512, flat.numel()),), generator=probe)
                arrays[f'{kind}.{tag}.gidx.{nm}'] = gi
                arrays[f'{kind}.{tag}.gval.{nm}'] = flat[gi]
                arrays[f'{kind}.{tag}.gstat.{nm}'] = np.array([flat.norm().item(), flat.abs().max().item()])
    save('sr_heads', 2e-5, **arrays)


if __name__ == '__main__':
    only = sys.argv[1:]
    gens = dict(bias_act=gen_bias_act, upfirdn2d=gen_upfirdn2d, filtered_lrelu=gen_filtered_lrelu, conv=gen_conv2d_resample, renderer=gen_renderer,
                graph_small=gen_graph_small, graph_full=gen_graph_full, loss=gen_loss_glue, projector_loop=gen_projector_loop, tuner_loop=gen_tuner_loop,
                inference=gen_inference, pose_net=gen_pose_net, e4e=gen_e4e, sr_heads=gen_sr_heads, c3_full=gen_c3_full, c2_full=gen_c2_full,
                c4_full=gen_c4_full, graph_full_cond=gen_graph_full_cond)
    mpath = os.path.join(HERE, 'MANIFEST.json')
    if only and os.path.exists(mpath):
        MANIFEST.update(json.load(open(mpath)).get('fixtures', {}))
    for k, fn in gens.items():
        if not only or k in only:
            fn()
    json.dump(dict(generator='tests/golden/make_golden.py', reference='cvlab-kaist/3DGAN-Inversion @ /root/reference (CPU, *_ref op path)',
                   torch=torch.__version__, dtype='float32', fixtures=MANIFEST), open(mpath, 'w'), indent=1, sort_keys=True)
    print('ORACLE PINNED against the reference on all generated cases.')
