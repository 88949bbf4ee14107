"""The product's explicit impl='ref' of the three L1 operators (inv3d_amd/torch_utils/ops/_ref_impl.py) against the fixtures recorded from the
reference's own `_ref` implementations (tests/golden/{bias_act,upfirdn2d,filtered_lrelu}.npz; torch_utils/ops/bias_act.py:84-88,
upfirdn2d.py:160-164, filtered_lrelu.py:113-120).  CPU only; does not import oracle/.  Also: the default impl='cuda' still refuses CPU tensors
(no automatic fallback)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def t(a):
    return torch.from_numpy(np.asarray(a)).clone()


def close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol * scale, f'err {err:.3e} > {tol} * {scale:.3e}'


def test_bias_act_ref_fixture():
    from inv3d_amd.torch_utils.ops import bias_act as B
    d = np.load(os.path.join(GOLD, 'bias_act.npz'), allow_pickle=False)
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        dim, clamp, gain, alpha = d[f'{k}_meta']
        x = t(d[f'{k}_x']).requires_grad_(True)
        b = t(d[f'{k}_b']).requires_grad_(True)
        y = B.bias_act(x, b, dim=int(dim), act=str(d[f'{k}_act']), alpha=None if alpha < 0 else float(alpha), gain=None if gain < 0 else float(gain),
                       clamp=None if clamp < 0 else float(clamp), impl='ref')
        close(y, t(d[f'{k}_y']), 1e-6)
        dx, db = torch.autograd.grad(y, [x, b], t(d[f'{k}_dy']))
        close(dx, t(d[f'{k}_dx']), 1e-6)
        close(db, t(d[f'{k}_db']), 1e-5)


def test_upfirdn2d_ref_fixture():
    from inv3d_amd.torch_utils.ops import upfirdn2d as U
    d = np.load(os.path.join(GOLD, 'upfirdn2d.npz'), allow_pickle=False)
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = d[f'{k}_meta']
        f = t(d[f'{k}_f'])
        f = None if f.numel() == 0 else f
        x = t(d[f'{k}_x']).requires_grad_(True)
        y = U.upfirdn2d(x, f, up=(int(m[0]), int(m[1])), down=(int(m[2]), int(m[3])), padding=[int(v) for v in m[4:8]], flip_filter=bool(m[8]), gain=float(m[9]),
                        impl='ref')
        close(y, t(d[f'{k}_y']), 1e-6)
        dx, = torch.autograd.grad(y, x, t(d[f'{k}_dy']))
        close(dx, t(d[f'{k}_dx']), 1e-6)
    x, f44 = t(d['w_x']), t(d['f44'])
    close(U.upsample2d(x, f44, impl='ref'), t(d['w_upsample2d_y']), 1e-6)
    close(U.downsample2d(x, f44, impl='ref'), t(d['w_downsample2d_y']), 1e-6)
    close(U.filter2d(x, f44, impl='ref'), t(d['w_filter2d_y']), 1e-6)


def test_filtered_lrelu_ref_fixture():
    from inv3d_amd.torch_utils.ops import filtered_lrelu as FL
    d = np.load(os.path.join(GOLD, 'filtered_lrelu.npz'), allow_pickle=False)
    opt = lambda a: None if a.size == 0 else t(a)       # noqa: E731
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = d[f'{k}_meta']
        kw = dict(fu=opt(d[f'{k}_fu']), fd=opt(d[f'{k}_fd']), up=int(m[0]), down=int(m[1]), padding=[int(v) for v in m[2:6]], gain=float(m[6]),
                  slope=float(m[7]), clamp=None if m[8] < 0 else float(m[8]), flip_filter=bool(m[9]))
        b = opt(d[f'{k}_b'])
        x = t(d[f'{k}_x']).requires_grad_(True)
        if b is not None:
            b = b.requires_grad_(True)
        y = FL.filtered_lrelu(x, b=b, impl='ref', **kw)
        close(y, t(d[f'{k}_y']), 1e-6)
        g = torch.autograd.grad(y, [x] + ([b] if b is not None else []), t(d[f'{k}_dy']))
        close(g[0], t(d[f'{k}_dx']), 1e-5)
        if b is not None:
            close(g[1], t(d[f'{k}_db']), 1e-5)


def test_default_impl_has_no_cpu_fallback():
    """impl='cuda' (the default) on CPU tensors raises -- the reference falls back to `_ref` silently (bias_act.py:86-88); the product does not."""
    from inv3d_amd.torch_utils.ops import bias_act as B, upfirdn2d as U, filtered_lrelu as FL
    x = torch.randn(1, 4, 8, 8)
    for call in (lambda: B.bias_act(x, torch.zeros(4)), lambda: U.upfirdn2d(x, torch.ones(4)), lambda: FL.filtered_lrelu(x, fu=torch.ones(4), fd=torch.ones(4), up=2, down=2)):
        with pytest.raises(Exception):
            call()


def test_ref_impl_does_not_import_the_oracle():
    import re
    from inv3d_amd.torch_utils.ops import _ref_impl
    src = open(_ref_impl.__file__).read()
    assert not re.search(r'^\s*(import|from)\s+oracle\b', src, re.M)
