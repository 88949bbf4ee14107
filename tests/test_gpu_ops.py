"""GPU parity tests (run with `-m gpu` on the MI355X box): every HIP entry point, through the C-ABI, against
(a) the golden vectors recorded from the imported reference and (b) the CPU oracle on seeded inputs.
Tolerances: fp32 op level 1e-5; reductions over many elements get a looser relative bound."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eg3d_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def t(a, dev=DEV):
    return torch.from_numpy(np.asarray(a)).to(dev)


def close(a, b, tol, what=''):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), f'{what}: non-finite values'
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol * scale, f'{what}: err {err:.3e} > {tol} * {scale:.3e}'


def close_rel(a, b, tol, what=''):
    """max |a - b| <= tol x max |b| (no floor at 1: gradient-sized operands)."""
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape and torch.isfinite(a).all(), what
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= tol * scale, f'{what}: err {err:.3e} > {tol} * {scale:.3e}'


def close_most(a, b, tol, what='', frac=0.002, loose=0.05):
    """Per-ray coordinate gradients are piecewise constant in the sample position (bilinear texel boundaries): a sample
    whose coordinate differs by one ulp between CPU and GPU can flip a floor() and change that ray's gradient by O(1/samples).
    Require the tight tolerance on all but `frac` of the rows (at least one) and a loose bound on the rest.  Rounds 1-3 allowed 2 % of the
    rays; since the tie order of unify_samples follows the reference's sort (round 4) every case of this suite has ZERO rows above the
    tolerance and a worst row of 4e-6 -- the allowance is now 0.2 % (one ray of 200, four of 2048), for the 3-per-million importance
    samples whose uniform sits within 4e-6 of a CDF edge (test_sampler_indices_exact counts those)."""
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape and torch.isfinite(a).all(), what
    scale = max(1.0, float(b.abs().max()))
    err = (a - b).abs().reshape(-1, a.shape[-1]).amax(-1)
    bad = int((err > tol * scale).sum())
    assert bad <= max(1, int(frac * err.numel())), f'{what}: {bad}/{err.numel()} rows above {tol}'
    assert float(err.max()) <= loose * scale, f'{what}: worst row {float(err.max()):.3e}'


@pytest.fixture(scope='module')
def ops():
    from inv3d_amd.torch_utils.ops import bias_act, upfirdn2d, conv2d_resample
    return dict(bias_act=bias_act, upfirdn2d=upfirdn2d, conv2d_resample=conv2d_resample)


# ------------------------------------------------------------------------------------------------- bias_act
def test_bias_act_golden(golden, ops):
    d = golden('bias_act')
    ba = ops['bias_act']
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        dim, clamp, gain, alpha = d[f'{k}_meta']
        act = str(d[f'{k}_act'])
        kw = dict(dim=int(dim), act=act, alpha=None if alpha < 0 else float(alpha), gain=None if gain < 0 else float(gain),
                  clamp=None if clamp < 0 else float(clamp))
        x = t(d[f'{k}_x']).requires_grad_(True)
        b = t(d[f'{k}_b']).requires_grad_(True)
        y = ba.bias_act(x, b, **kw)
        close(y, d[f'{k}_y'], 1e-5, f'bias_act {act} fwd')
        dx, db = torch.autograd.grad(y, [x, b], t(d[f'{k}_dy']))
        close(dx, d[f'{k}_dx'], 1e-5, f'bias_act {act} dx')
        close(db, d[f'{k}_db'], 1e-4, f'bias_act {act} db')


@pytest.mark.parametrize('act', ['tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish', 'lrelu'])
def test_bias_act_second_order(ops, act):
    """grad=2 kernel vs double-backward of the oracle."""
    ba = ops['bias_act']
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 5, 5, generator=g)
    b = torch.randn(8, generator=g)
    dy = torch.randn(2, 8, 5, 5, generator=g)
    v = torch.randn(2, 8, 5, 5, generator=g)

    def run(fn, dev):
        xx, bb = x.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        dyy = dy.to(dev).requires_grad_(True)
        y = fn(xx, bb)
        dx, = torch.autograd.grad(y, xx, dyy, create_graph=True)
        gx, gdy = torch.autograd.grad(dx, [xx, dyy], v.to(dev), allow_unused=True)
        return dx, (gx if gx is not None else torch.zeros_like(xx)), gdy
    ref = run(lambda a, c: O.bias_act(a, c, act=act, gain=1.3), 'cpu')
    got = run(lambda a, c: ba.bias_act(a, c, act=act, gain=1.3), DEV)
    for r, h, nm in zip(ref, got, ('dx', 'd2x', 'd_dy')):
        close(h, r, 2e-5, f'{act} {nm}')


@pytest.mark.parametrize('dtype', [torch.float16, torch.float64])
@pytest.mark.parametrize('cl', [False, True])
def test_bias_act_dtypes_layouts(ops, dtype, cl):
    ba = ops['bias_act']
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 6, 7, 5, generator=g)
    b = torch.randn(6, generator=g)
    ref = O.bias_act(x.double(), b.double(), act='lrelu', clamp=1.5)
    xg = x.to(DEV, dtype)
    if cl:
        xg = xg.contiguous(memory_format=torch.channels_last)
    y = ba.bias_act(xg, b.to(DEV, dtype), act='lrelu', clamp=1.5)
    assert y.dtype == dtype and y.stride() == xg.stride()
    close(y.double(), ref, 2e-3 if dtype == torch.float16 else 1e-7, 'bias_act dtype')   # alpha/gain/clamp are C floats, as in the reference ABI


def test_bias_act_large_vectorised(ops):
    ba = ops['bias_act']
    x = torch.randn(1, 128, 96, 96, device=DEV)
    b = torch.randn(128, device=DEV)
    for xx in (x, x.contiguous(memory_format=torch.channels_last)):
        y = ba.bias_act(xx, b, act='lrelu', clamp=256)
        ref = O.bias_act(xx.cpu(), b.cpu(), act='lrelu', clamp=256)
        close(y, ref, 1e-6, 'bias_act large')


# ------------------------------------------------------------------------------------------------- upfirdn2d
def test_upfirdn2d_golden(golden, ops):
    d = golden('upfirdn2d')
    up = ops['upfirdn2d']
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = d[f'{k}_meta']
        f = t(d[f'{k}_f'])
        f = None if f.numel() == 0 else f
        for cl in (False, True):
            x = t(d[f'{k}_x'])
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
            x = x.requires_grad_(True)
            y = up.upfirdn2d(x, f, up=[int(m[0]), int(m[1])], down=[int(m[2]), int(m[3])], padding=[int(v) for v in m[4:8]],
                             flip_filter=bool(m[8]), gain=float(m[9]))
            close(y, d[f'{k}_y'], 1e-5, f'upfirdn2d case {i} cl={cl}')
            dx, = torch.autograd.grad(y, x, t(d[f'{k}_dy']))
            close(dx, d[f'{k}_dx'], 1e-5, f'upfirdn2d case {i} dx')
    x, f44 = t(d['w_x']), t(d['f44'])
    close(up.upsample2d(x, f44), d['w_upsample2d_y'], 1e-5)
    close(up.downsample2d(x, f44), d['w_downsample2d_y'], 1e-5)
    close(up.filter2d(x, f44), d['w_filter2d_y'], 1e-5)
    close(up.setup_filter([1, 3, 3, 1]), d['f44'], 1e-7)


def test_filtered_lrelu_golden(golden):
    """eg3d_filtered_lrelu (one fused launch) vs the reference's _filtered_lrelu_ref fixtures: forward, dx, db, fp16, and a
    second-order check (gradient of the gradient is the same masked-FIR operator, transposed twice)."""
    from inv3d_amd.torch_utils.ops import filtered_lrelu as FL
    d = golden('filtered_lrelu')
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = d[f'{k}_meta']
        opt = lambda a: None if a.size == 0 else t(a)
        kw = dict(fu=opt(d[f'{k}_fu']), fd=opt(d[f'{k}_fd']), up=int(m[0]), down=int(m[1]), padding=[int(v) for v in m[2:6]], gain=float(m[6]),
                  slope=float(m[7]), clamp=None if m[8] < 0 else float(m[8]), flip_filter=bool(m[9]))
        x = t(d[f'{k}_x']).requires_grad_(True)
        b = opt(d[f'{k}_b'])
        if b is not None:
            b = b.requires_grad_(True)
        y = FL.filtered_lrelu(x, b=b, **kw)
        close(y, d[f'{k}_y'], 1e-5, f'filtered_lrelu case {i}')
        dy = t(d[f'{k}_dy'])
        g = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dy, create_graph=True)
        close(g[0], d[f'{k}_dx'], 2e-5, f'filtered_lrelu case {i} dx')
        if b is not None:
            close(g[1], d[f'{k}_db'], 5e-5, f'filtered_lrelu case {i} db')
        # second order: d/d(dy) <dx, v> must equal the forward linearisation applied to v  (oracle autograd as the checker)
        v = torch.randn_like(x)
        xo = t(d[f'{k}_x']).cpu().requires_grad_(True)
        kwo = {a: (b_.cpu() if torch.is_tensor(b_) else b_) for a, b_ in kw.items()}
        yo = O.filtered_lrelu(xo, b=None if b is None else b.detach().cpu(), **kwo)
        dyo = dy.cpu().requires_grad_(True)
        gxo, = torch.autograd.grad(yo, xo, dyo, create_graph=True)
        ggo, = torch.autograd.grad(gxo, dyo, v.cpu())
        dyg = dy.clone().requires_grad_(True)
        y2 = FL.filtered_lrelu(x, b=b, **kw)
        gx, = torch.autograd.grad(y2, x, dyg, create_graph=True)
        gg, = torch.autograd.grad(gx, dyg, v)
        close(gg, ggo, 5e-5, f'filtered_lrelu case {i} second order')
        # fp16 storage
        yh = FL.filtered_lrelu(x.detach().half(), b=None if b is None else b.detach().half(), **kw)
        assert yh.dtype == torch.float16
        close(yh.float(), d[f'{k}_y'], 2e-2, f'filtered_lrelu case {i} fp16')
    # the explicit impl='ref' (the reference's own keyword, filtered_lrelu.py:113-120): the product's plain-torch composite agrees with the kernel
    with torch.no_grad():       # (x, b, kw: the last fixture case)
        close(FL.filtered_lrelu(x.detach(), b=None if b is None else b.detach(), impl='ref', **kw), y.detach(), 2e-5, 'filtered_lrelu impl=ref vs kernel')


def test_upfirdn2d_nhwc_fused(golden):
    """The channels-last float4 resampler used on the fused path vs the oracle (skip upsample, its adjoint, FIR adjoint)."""
    from inv3d_amd import hipops as H
    g = torch.Generator().manual_seed(5)
    f = O.setup_filter([1, 3, 3, 1])
    x = torch.randn(2, 8, 9, 7, generator=g)
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    close(H.upfirdn2d_nhwc(xc, f.to(DEV), up=2, pad=(2, 1, 2, 1), gain=4.0), O.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], gain=4.0), 1e-5)
    close(H.upfirdn2d_nhwc(xc, f.to(DEV), down=2, pad=(1, 1, 1, 1), flip=True, gain=4.0),
          O.upfirdn2d(x, f, down=2, padding=[1, 1, 1, 1], flip_filter=True, gain=4.0), 1e-5)
    close(H.upfirdn2d_nhwc(xc, f.to(DEV), pad=(2, 2, 2, 2), flip=True, gain=4.0), O.upfirdn2d(x, f, padding=[2, 2, 2, 2], flip_filter=True, gain=4.0), 1e-5)


# ------------------------------------------------------------------------------------------------- conv
def test_conv2d_resample_golden(golden, ops):
    d = golden('conv2d_resample')
    c2r = ops['conv2d_resample']
    f44 = t(d['f44'])
    ran = 0
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = [int(v) for v in d[f'{k}_meta']]
        x = t(d[f'{k}_x']).requires_grad_(True)
        w = t(d[f'{k}_w']).requires_grad_(True)
        kw = dict(f=f44, up=m[0], down=m[1], padding=m[2:6], groups=m[6], flip_weight=bool(m[7]))
        y = c2r.conv2d_resample(x, w, **kw)          # every branch of the reference incl. groups > 1 and strided k x k (round 3)
        ran += 1
        close(y, d[f'{k}_y'], 1e-5, f'conv2d_resample case {i}')
        dx, dw = torch.autograd.grad(y, [x, w], t(d[f'{k}_dy']))
        close(dx, d[f'{k}_dx'], 1e-5, f'conv2d_resample case {i} dx')
        close(dw, d[f'{k}_dw'], 1e-5, f'conv2d_resample case {i} dw')
    assert ran == int(d['ncases']) and ran >= 6


@pytest.mark.parametrize('shape', [(1, 32, 16, 16, 64, 1), (2, 64, 8, 8, 160, 1), (1, 128, 24, 24, 128, 3), (3, 16, 5, 7, 12, 3),
                                   (1, 512, 4, 4, 512, 3), (1, 256, 40, 40, 256, 3)])
@pytest.mark.parametrize('prec,tol', [('f32', 2e-5), ('bf16x6', 2e-5), ('f16x3', 2e-5), ('bf16x3', 3e-4)])
def test_conv_igemm_vs_torch(shape, prec, tol):
    """All tile configurations / split-K / matrix-core arithmetic modes of the implicit GEMM vs F.conv2d evaluated in fp64.
    'bf16x6' (the default: six bf16 products per fp32 product) must meet the SAME bound as the exact-fp32 MFMA path."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co, k = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=k // 2).float()
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wf = H.pack_weight_fwd(wt.to(DEV))
    cop = (co + 3) // 4 * 4
    for ks in (1, 3):
        out = H.zeros_cl(n, cop, h, w, DEV)
        H.conv_igemm(xc, wf, ci, co, out, H.classes_corr(h, w, k, k, k // 2), epi=L.EPI_ATOMIC if ks > 1 else L.EPI_STORE, ksplit=ks,
                     precision=prec)
        close(out[:, :co], ref, tol, f'conv_igemm {shape} ksplit {ks} {prec}')


def test_conv_igemm_f16x3_range_normalisation():
    """'f16x3' forms fp32 products from two fp16 pieces per operand: as accurate as the fp32 MFMA path for operands of ordinary
    magnitude, and -- given max|A| (what epilogue_bwd reports for a gradient tensor) -- for tiny / huge operands as well."""
    from inv3d_amd import hipops as H
    g = torch.Generator().manual_seed(9)
    n, ci, h, co = 1, 256, 16, 128
    w = (torch.randn(co, ci, 3, 3, generator=g) / 48).to(DEV)
    wf = H.pack_weight_fwd(w)
    cls = H.classes_corr(h, h, 3, 3, 1)
    for scale in (1.0, 3e-7, 2.5e6):
        x = (torch.randn(n, ci, h, h, generator=g) * scale)
        ref = torch.nn.functional.conv2d(x.double(), w.double().cpu(), padding=1)
        xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
        amax = xc.abs().max().reshape(1)
        err = {}
        for name, kw in (('f32', dict(precision='f32')), ('f16x3', dict(precision='f16x3', a_amax=amax)),
                         ('f16x3 with a 4x loose bound', dict(precision='f16x3', a_amax=amax / 4, a_amax_mul=16.0))):
            out = H.zeros_cl(n, co, h, h, DEV)
            H.conv_igemm(xc, wf, ci, co, out, cls, **kw)
            err[name] = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err['f16x3'] <= 2.0 * err['f32'] + 1e-7 and err['f16x3 with a 4x loose bound'] <= 4.0 * err['f32'] + 1e-7, (scale, err)
    # the producer side: epilogue_bwd reports max|dz|
    dout = torch.randn(1, 32, 8, 8, generator=g).to(DEV).contiguous(memory_format=torch.channels_last) * 1e-5
    outv = torch.randn(1, 32, 8, 8, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    dz, am = torch.empty_like(dout), torch.zeros(1, device=DEV)
    H.epilogue_bwd(dout, outv, dz, act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, dz_amax=am)
    assert float(am) == float(dz.abs().max()) > 0


def test_conv_igemm_split_bf16_is_fp32_equivalent():
    """The default arithmetic (bf16x6) is as close to the fp64 result as the exact-fp32 MFMA path on a long reduction with badly
    scaled operands (values spread over 12 orders of magnitude), where a reduced-precision product would show."""
    from inv3d_amd import hipops as H
    g = torch.Generator().manual_seed(8)
    n, ci, h, co = 1, 512, 16, 128
    x = torch.randn(n, ci, h, h, generator=g) * torch.exp(torch.randn(n, ci, 1, 1, generator=g) * 4)
    wt = torch.randn(co, ci, 3, 3, generator=g) * torch.exp(torch.randn(1, ci, 1, 1, generator=g) * 4) / 70
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wf = H.pack_weight_fwd(wt.to(DEV))
    err = {}
    for prec in ('f32', 'bf16x6', 'bf16x3'):
        out = H.zeros_cl(n, co, h, h, DEV)
        H.conv_igemm(xc, wf, ci, co, out, H.classes_corr(h, h, 3, 3, 1), precision=prec)
        err[prec] = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err['bf16x6'] <= 2.0 * err['f32'] + 1e-7, err
    assert err['f32'] < 1e-5 and err['bf16x3'] < 1e-3, err


def _layer_case(n, ci, co, res, up, seed, noise_kind='const', clamp=None):
    g = torch.Generator().manual_seed(seed)
    hin = res // up
    P = {
        'L.weight': torch.randn(co, ci, 3, 3, generator=g),
        'L.bias': torch.randn(co, generator=g) * 0.1,
        'L.affine.weight': torch.randn(ci, 32, generator=g),
        'L.affine.bias': torch.ones(ci),
        'L.noise_strength': torch.tensor(0.07),
        'L.noise_const': torch.randn(res, res, generator=g),
        'L.resample_filter': O.setup_filter([1, 3, 3, 1]),
    }
    x = torch.randn(n, ci, hin, hin, generator=g)
    w = torch.randn(n, 32, generator=g)
    dy = torch.randn(n, co, res, res, generator=g)
    noise = torch.randn(n, 1, res, res, generator=g) if noise_kind == 'random' else None
    return P, x, w, dy, noise


@pytest.mark.parametrize('cfg', [(2, 16, 24, 8, 1, 'const', None), (2, 16, 24, 16, 2, 'const', None), (1, 32, 16, 16, 2, 'random', 0.9),
                                 (1, 64, 128, 32, 1, 'none', 256.0), (1, 128, 128, 64, 2, 'const', 256.0), (3, 8, 8, 4, 1, 'const', None),
                                 (1, 512, 512, 8, 2, 'const', None)])
def test_synthesis_layer_fwd_bwd(cfg):
    """Fused SynthesisLayer (conv + demod + noise + bias + lrelu + clamp) vs the oracle, all gradients."""
    from inv3d_amd.training.networks_stylegan2 import SynthesisLayer
    n, ci, co, res, up, noise_kind, clamp = cfg
    P, x, w, dy, noise = _layer_case(n, ci, co, res, up, seed=(n * 131 + ci * 17 + co * 7 + res * 3 + up), noise_kind=noise_kind, clamp=clamp)
    # oracle
    names = ['L.weight', 'L.bias', 'L.affine.weight', 'L.affine.bias', 'L.noise_strength', 'L.noise_const']
    Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = O.synthesis_layer(Pg, 'L', xr, wr, up, noise_kind, noise, clamp)
    gr = torch.autograd.grad(yr, [xr, wr] + [Pg[k] for k in names], dy, allow_unused=True)
    # hip
    layer = SynthesisLayer(ci, co, w_dim=32, resolution=res, up=up, conv_clamp=clamp).to(DEV)
    layer.load_state_dict({k[2:]: v for k, v in P.items()})
    layer.noise_const.requires_grad_(True)
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    yg = layer(xg, wg, noise_mode=noise_kind, noise_inject=noise.to(DEV) if noise is not None else None)
    close(yg, yr, 2e-5, f'layer fwd {cfg}')
    params = [layer.weight, layer.bias, layer.affine.weight, layer.affine.bias, layer.noise_strength, layer.noise_const]
    gg = torch.autograd.grad(yg, [xg, wg] + params, dy.to(DEV), allow_unused=True)
    for nm, a, b in zip(['x', 'w'] + names, gg, gr):
        if b is None:
            assert a is None or float(a.abs().max()) == 0, nm
            continue
        close(a, b, 1e-4, f'layer grad {nm} {cfg}')


@pytest.mark.parametrize('cfg', [(2, 16, 96, 8, None, True), (1, 32, 3, 16, 256.0, True), (2, 8, 3, 4, 0.5, False),
                                 # the low-latency launch of csrc/torgb_small.hip (>= 32 input channels, outputs a multiple of 32, few pixels):
                                 (2, 64, 96, 8, None, True), (1, 512, 96, 4, None, False), (1, 64, 96, 16, 0.5, True), (3, 136, 32, 7, None, True),
                                 # >= 8192 pixels, 96 outputs: the streaming form (torgb_mid_kernel / torgb_mid_bwd_kernel): K split over wave pairs (few tiles,
                                 # C % 64 == 0), whole contraction per wave (C = 96; 1152 tiles), two images, clamp
                                 (1, 128, 96, 96, None, True), (1, 96, 96, 96, None, False), (1, 32, 96, 192, None, True), (2, 64, 96, 80, None, True),
                                 (1, 256, 96, 96, 0.5, True),
                                 # four outputs, > 4096 pixels: the stream kernel of the SR head's toRGB (eg3d_torgb4_fwd)
                                 (1, 64, 3, 72, 2.0, True), (2, 128, 3, 48, None, False), (1, 256, 3, 68, 256.0, True), (1, 32, 4, 80, 0.7, False)])
def test_torgb_fwd_bwd(cfg):
    from inv3d_amd.training.networks_stylegan2 import ToRGBLayer
    n, ci, co, res, clamp, with_skip = cfg
    g = torch.Generator().manual_seed(9)
    P = {'T.weight': torch.randn(co, ci, 1, 1, generator=g), 'T.bias': torch.randn(co, generator=g) * 0.1,
         'T.affine.weight': torch.randn(ci, 32, generator=g), 'T.affine.bias': torch.ones(ci)}
    x = torch.randn(n, ci, res, res, generator=g)
    w = torch.randn(n, 32, generator=g)
    cp = (co + 3) // 4 * 4
    skip = torch.randn(n, cp, res, res, generator=g) if with_skip else None
    if skip is not None and cp != co:
        skip[:, co:] = 0
    dy = torch.randn(n, cp, res, res, generator=g)
    names = list(P)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = O.torgb_layer(Pg, 'T', xr, wr, clamp)
    if skip is not None:
        sr = skip.clone().requires_grad_(True)
        yr = yr + sr[:, :co]
    gr = torch.autograd.grad(yr, [xr, wr] + [Pg[k] for k in names], dy[:, :co])
    layer = ToRGBLayer(ci, co, w_dim=32, conv_clamp=clamp).to(DEV)
    layer.load_state_dict({k[2:]: v for k, v in P.items()})
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    sg = skip.to(DEV).requires_grad_(True) if skip is not None else None
    yg = layer(xg, wg, skip=sg)
    close(yg[:, :co], yr, 2e-5, f'torgb fwd {cfg}')
    dyg = dy.to(DEV).clone()
    dyg[:, co:] = 0
    gg = torch.autograd.grad(yg, [xg, wg, layer.weight, layer.bias, layer.affine.weight, layer.affine.bias] + ([sg] if sg is not None else []), dyg)
    for nm, a, b in zip(['x', 'w'] + names, gg, gr):
        close(a, b, 1e-4, f'torgb grad {nm} {cfg}')
    if sg is not None:
        close(gg[-1][:, :co], dy[:, :co], 0, 'torgb dskip')


@pytest.mark.parametrize('cfg', [(1, 16, 96, 8, None), (2, 32, 96, 16, None), (1, 16, 3, 32, None), (2, 16, 96, 6, None), (1, 16, 3, 16, 256.0),
                                 (1, 512, 96, 8, None), (2, 64, 96, 6, None), (1, 64, 96, 8, 2.0), (1, 256, 96, 64, None),
                                 (1, 128, 96, 96, None), (2, 64, 96, 128, None), (1, 96, 96, 160, None),          # the streaming form (torgb_mid_kernel)
                                 (1, 64, 3, 72, 1.5), (2, 128, 3, 66, None), (1, 256, 3, 70, 256.0)])
def test_torgb_takes_the_skip_image_at_half_resolution(cfg):
    """skip + toRGB of a 'skip' block (networks_stylegan2.py:433-436: img = upsample2d(img); img = img.add_(y)) with the up-sampling done
    inside the conv's epilogue (eg3d_conv_params::addend_up2): against the oracle's upfirdn2d on the CPU, values and every gradient.  The
    6 x 6, N = 2 case and the clamped case cannot take the fused form (tiles across images / clamp mask): they must fall back, same results."""
    from inv3d_amd.training.networks_stylegan2 import ToRGBLayer
    n, ci, co, res, clamp = cfg
    g = torch.Generator().manual_seed(19)
    P = {'T.weight': torch.randn(co, ci, 1, 1, generator=g), 'T.bias': torch.randn(co, generator=g) * 0.1,
         'T.affine.weight': torch.randn(ci, 32, generator=g), 'T.affine.bias': torch.ones(ci)}
    x = torch.randn(n, ci, res, res, generator=g)
    w = torch.randn(n, 32, generator=g)
    cp = (co + 3) // 4 * 4
    low = torch.randn(n, cp, res // 2, res // 2, generator=g)
    low[:, co:] = 0
    dy = torch.randn(n, cp, res, res, generator=g)
    dy[:, co:] = 0
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr, wr, lr_ = x.clone().requires_grad_(True), w.clone().requires_grad_(True), low.clone().requires_grad_(True)
    f = O.setup_filter([1, 3, 3, 1])
    yr = O.torgb_layer(Pg, 'T', xr, wr, clamp) + O.upfirdn2d(lr_, f, up=2, padding=[2, 1, 2, 1], gain=4.0)[:, :co]
    gr = torch.autograd.grad(yr, [xr, wr, lr_], dy[:, :co])
    layer = ToRGBLayer(ci, co, w_dim=32, conv_clamp=clamp).to(DEV)
    layer.load_state_dict({k[2:]: v for k, v in P.items()})
    xg, wg, lg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), low.to(DEV).requires_grad_(True)
    yg = layer(xg, wg, skip=lg, skip_up=True)
    close(yg[:, :co], yr, 2e-5, f'torgb + up-sampled skip fwd {cfg}')
    gg = torch.autograd.grad(yg, [xg, wg, lg], dy.to(DEV))
    for nm, a, b in zip(['x', 'w', 'skip'], gg, gr):
        close(a[:, :co] if nm == 'skip' else a, b[:, :co] if nm == 'skip' else b, 1e-4, f'torgb + up-sampled skip grad {nm} {cfg}')


@pytest.mark.parametrize('case', [('conv', 2, 2, 1), ('conv', 3, 3, 2), ('conv', 2, 0, 4), ('convT', 2, 1, 1), ('convT', 3, 1, 2), ('convT', 2, 0, 1), ('convT', 3, 0, 2)])
def test_conv2d_gradfix_dilation_and_output_padding(case):
    """The two corners of F.conv2d / F.conv_transpose2d that conv2d_gradfix hands to ATen unchanged (torch_utils/ops/conv2d_gradfix.py:37-45)
    and that raised until round 3: dilated stride-1 k x k kernels, and output_padding of a transposed conv -- values and both gradients
    against torch in float64, with groups."""
    from inv3d_amd.torch_utils.ops import conv2d_gradfix
    kind, a, b, groups = case
    g = torch.Generator().manual_seed(21)
    ci, co, k = 8, 12, 3
    x = torch.randn(2, ci, 11, 9, generator=g)
    dyn = None
    if kind == 'conv':                     # a = dilation, b = padding
        w = torch.randn(co, ci // groups, k, k, generator=g) / 5
        ref = lambda xx, ww: torch.nn.functional.conv2d(xx, ww, None, 1, b, a, groups)
        mine = lambda xx, ww: conv2d_gradfix.conv2d(xx, ww, None, 1, b, a, groups)
    else:                                  # a = stride, b = padding, output_padding = a - 1
        w = torch.randn(ci, co // groups, k, k, generator=g) / 5
        ref = lambda xx, ww: torch.nn.functional.conv_transpose2d(xx, ww, None, a, b, a - 1, groups)
        mine = lambda xx, ww: conv2d_gradfix.conv_transpose2d(xx, ww, None, a, b, a - 1, groups)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = ref(xr, wr)
    dy = torch.randn(yr.shape, generator=g).double()
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], dy)
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    yg = mine(xg, wg)
    close(yg, yr, 2e-5, f'{case} forward')
    gx, gw = torch.autograd.grad(yg, [xg, wg], dy.float().to(DEV))
    close(gx, gxr, 2e-5, f'{case} dx')
    close(gw, gwr, 5e-5, f'{case} dw')


@pytest.mark.parametrize('n,trainable', [(1, False), (3, False), (2, True)])
def test_style_bank_matches_torch(n, trainable):
    """fused.StyleBankFn (eg3d_style_affine_fwd / _bwd: the per-layer affines `styles = affine(w) * gain` of networks_stylegan2.py:98-108,129-137 and
    the demodulation coefficients d = rsqrt(sum_i (w_oi s_i)^2 + 1e-8) of :62-67, all layers in one launch per direction) against the same
    expressions in float64 autograd: styles, d, d ws and -- with trainable affines -- their weight / bias gradients.  Layers of different widths,
    ws rows and post scales, with and without bias / demodulation."""
    from inv3d_amd import fused
    g = torch.Generator().manual_seed(13)
    D, Lw = 64, 5
    specs = [(96, 0, 1.0, True, 40), (33, 2, 0.37, True, None), (512, 4, 1.0, False, 16), (8, 4, 2.0, True, 24), (130, 1, 1.0, True, 7)]    # (C, ws row, post, bias, Co | None)
    ws = torch.randn(n, Lw, D, generator=g)
    Ws = [torch.randn(c, D, generator=g) for c, *_ in specs]
    Bs = [torch.randn(c, generator=g) if hb else None for c, _, _, hb, _ in specs]
    Wq = [torch.rand(co, c, generator=g) + 0.05 if co else None for c, _, _, _, co in specs]          # wsq = sum over taps of w^2 >= 0
    wgain, bgain = 1.0 / math.sqrt(D), 1.0
    # float64 reference
    wr = ws.double().requires_grad_(True)
    Wr = [w.double().requires_grad_(True) for w in Ws]
    Br = [b.double().requires_grad_(True) if b is not None else None for b in Bs]
    outs_r, ds_r = [], []
    for (c, row, post, hb, co), w, b, q in zip(specs, Wr, Br, Wq):
        st = (wr[:, row] @ (w * wgain).t() + (b * bgain if b is not None else 0.0)) * post
        outs_r.append(st)
        ds_r.append(torch.rsqrt((st.square().unsqueeze(1) * q.double().unsqueeze(0)).sum(2) + 1e-8) if q is not None else None)
    gs = [torch.randn(o.shape, generator=g).double() for o in outs_r]
    gd = [torch.randn(d.shape, generator=g).double() if d is not None else None for d in ds_r]
    loss = sum((o * a).sum() for o, a in zip(outs_r, gs)) + sum((d * a).sum() for d, a in zip(ds_r, gd) if d is not None)
    leaves = [wr] + (Wr + [b for b in Br if b is not None] if trainable else [])
    grads_r = torch.autograd.grad(loss, leaves)
    # product
    wg = ws.to(DEV).requires_grad_(True)
    Wg = [w.to(DEV).requires_grad_(trainable) for w in Ws]
    Bg = [b.to(DEV).requires_grad_(trainable) if b is not None else None for b in Bs]
    plan, params = [], []
    for (c, row, post, hb, co), w, b, q in zip(specs, Wg, Bg, Wq):
        plan.append((row, wgain, bgain, post, hb, q.to(DEV).contiguous() if q is not None else None))
        params.append(w)
        if hb:
            params.append(b)
    res = fused.StyleBankFn.apply(wg, tuple(plan), *params)
    outs, rest = res[:len(specs)], list(res[len(specs):])
    loss_g, di = 0.0, 0
    for i, (o, ref) in enumerate(zip(outs, outs_r)):
        close(o, ref.float(), 2e-5, f'styles of layer {i}')
        loss_g = loss_g + (o * gs[i].float().to(DEV)).sum()
    for i, ref in enumerate(ds_r):
        if ref is not None:
            close(rest[di], ref.float(), 2e-5, f'demodulation of layer {i}')
            loss_g = loss_g + (rest[di] * gd[i].float().to(DEV)).sum()
            di += 1
    leaves_g = [wg] + (Wg + [b for b in Bg if b is not None] if trainable else [])
    grads = torch.autograd.grad(loss_g, leaves_g)
    names = ['ws'] + ([f'W{i}' for i in range(len(Wg))] + [f'b{i}' for i, b in enumerate(Bg) if b is not None] if trainable else [])
    for nm, a, b in zip(names, grads, grads_r):
        close(a, b.float(), 5e-5, f'style bank d {nm}')


# ------------------------------------------------------------------------------------------------- renderer
def test_ray_gen_golden(golden):
    from inv3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    d = golden('renderer')
    c2w = t(d['rs_c2w']).requires_grad_(True)
    K = t(d['rs_K']).requires_grad_(True)
    rs = RaySampler()
    o, dr = rs(c2w, K, 8)
    close(o, d['rs_o'], 1e-6, 'ray origins'); close(dr, d['rs_d'], 1e-6, 'ray dirs')
    g = torch.autograd.grad([o, dr], [c2w, K], [t(d['rs_go']), t(d['rs_gd'])])
    close(g[0], d['rs_dc2w'], 1e-5, 'd cam2world'); close(g[1], d['rs_dK'], 1e-5, 'd intrinsics')
    close(rs.calculate_xyz_of_depth(o[:1], dr[:1], t(d['rs_depth'])[0]), d['rs_xyz'], 1e-6, 'xyz of depth')


def _decoder(P):
    from inv3d_amd.training.triplane import OSGDecoder
    dec = OSGDecoder(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32}).to(DEV)
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in P.items() if k.startswith('decoder.')})
    return dec


def test_render_golden(golden):
    """ImportanceRenderer.forward (D = 12+12) vs the reference's outputs and gradients (planes, cam2world, decoder)."""
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from inv3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
    d = golden('renderer')
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=5)
    dec = _decoder(P)
    planes = t(d['rn_planes']).requires_grad_(True)
    c2w = t(d['rn_c2w']).requires_grad_(True)
    o, dr = RaySampler()(c2w, t(d['rn_K']), 6)
    R = ImportanceRenderer()
    R.set_uniforms(t(d['rn_u1']), t(d['rn_u2']))
    rgb, dep, ws = R(planes, dec, o, dr, cfg.rendering)
    close(rgb, d['rn_rgb'], 1e-5, 'render rgb'); close(dep, d['rn_depth'], 1e-5, 'render depth'); close(ws, d['rn_wsum'], 1e-5, 'render wsum')
    params = [dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias]
    g = torch.autograd.grad([rgb, dep], [planes, c2w] + params, [t(d['rn_grgb']), t(d['rn_gdepth'])])
    close(g[0], d['rn_dplanes'], 5e-5, 'd planes'); close(g[1], d['rn_dc2w'], 1e-4, 'd cam2world')
    for a, k in zip(g[2:], ('rn_dw0', 'rn_db0', 'rn_dw1', 'rn_db1')):
        close(a, d[k], 1e-4, k)
    # 'auto' ray limits
    R.set_uniforms(t(d['rn_u1']), t(d['rn_u2']))
    rgb, dep, _ = R(planes, dec, o, dr, dict(cfg.rendering, ray_start='auto', ray_end='auto'))
    close(rgb, d['rn_auto_rgb'], 1e-5, 'render auto rgb'); close(dep, d['rn_auto_depth'], 1e-5, 'render auto depth')


@pytest.mark.parametrize('variant', ['ffhq48', 'uneven'])
def test_render_feature_row_path_equals_gather_in_decoder(variant):
    """eg3d_render_params.feat_rows (the tri-plane gather as its own pass, rows re-read by the decoder kernels in forward and backward)
    against the same renderer with the gather inside the decoder kernels: the interpolation is the same instruction sequence, so the
    forward is bit-identical; the gradients differ only by the summation order of atomically accumulated sums.  Also the opt-in
    three-product fp16 accumulation of the plane gradient (EG3D_SCATTER_F16) against the exact fp32 one."""
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from inv3d_amd import fused, hipops as H
    cfg = O.full_config()
    opts = dict(cfg.rendering)
    if variant == 'uneven':
        opts['depth_resolution'], opts['depth_resolution_importance'] = 40, 24
    res, n = 32, 2
    P = O.synth_params(O.small_config(), seed=7)
    g = torch.Generator().manual_seed(5)
    planes = (torch.randn(n, 3, 32, 64, 64, generator=g) * 0.8)
    cam = O.synth_cameras(n, seed=11)
    o, dr = O.ray_sampler(cam[:, :16].reshape(n, 4, 4), cam[:, 16:].reshape(n, 3, 3), res)
    dc, df = opts['depth_resolution'], opts['depth_resolution_importance']
    u1, u2 = torch.rand(n, res * res, dc, 1, generator=g), torch.rand(n * res * res, df, generator=g)
    g_rgb, g_dep = torch.randn(n, res * res, 32, generator=g).to(DEV), torch.randn(n, res * res, 1, generator=g).to(DEV)
    dec = _decoder(P)

    def run(feat, f16):
        old = fused.RENDER_FEAT_ROWS, H.SCATTER_F16
        fused.RENDER_FEAT_ROWS, H.SCATTER_F16 = feat, f16
        try:
            R = ImportanceRenderer()
            R.set_uniforms(u1.to(DEV), u2.to(DEV))
            pg = planes.to(DEV).requires_grad_(True)
            og, dg = o.to(DEV).requires_grad_(True), dr.to(DEV).requires_grad_(True)
            rgb, dep, ws = R(pg, dec, og, dg, opts)
            gg = torch.autograd.grad([rgb, dep], [pg, og, dg], [g_rgb, g_dep])
            torch.cuda.synchronize()
            return (rgb, dep, ws) + tuple(gg)
        finally:
            fused.RENDER_FEAT_ROWS, H.SCATTER_F16 = old
    a, b, c = run(False, False), run(True, False), run(True, True)
    for k in range(3):
        assert torch.equal(a[k], b[k]), f'forward output {k} differs between the gather paths'
    close(b[3], a[3], 2e-6, 'd planes'); close_most(b[4], a[4], 1e-5, 'd origins'); close_most(b[5], a[5], 1e-5, 'd dirs')
    close(c[3], b[3], 2e-6, 'd planes, fp16 three-product accumulation vs fp32')


def test_upfirdn2d_nhwc_with_addend():
    """img = upsample2d(img) + y of a clamped toRGB layer (networks_stylegan2.py:453-457) in one pass (eg3d_upfirdn2d_nhwc_add) against the
    up-sampling pass followed by an add: the same multiply-adds in the same order, so bit-identical; refusals for the forms it does not cover."""
    from inv3d_amd import hipops as H
    from inv3d_amd._lib import Eg3dHipError
    g = torch.Generator().manual_seed(3)
    f = torch.tensor([1., 3., 3., 1.])
    f2 = (f[:, None] * f[None, :] / 64).to(DEV)
    for n, c, h, w in [(1, 4, 32, 32), (2, 8, 9, 7)]:
        x = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        a = torch.randn(n, c, 2 * h, 2 * w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        sep = H.upfirdn2d_nhwc(x, f2, up=2, pad=(2, 1, 2, 1), gain=4.0) + a
        one = H.upfirdn2d_nhwc(x, f2, up=2, pad=(2, 1, 2, 1), gain=4.0, addend=a)
        assert torch.equal(one, sep)
    with pytest.raises(Eg3dHipError):         # the plain 4 x 4 FIR form (2 x 2 outputs per thread) has no addend path
        H.upfirdn2d_nhwc(x, f2, pad=(2, 1, 2, 1), addend=torch.zeros(2, 8, 9, 7, device=DEV).contiguous(memory_format=torch.channels_last))


@pytest.mark.parametrize('variant', ['ffhq48', 'wide_range', 'ragged'])
def test_decoder_weight_gradients_in_the_backward_kernel(variant):
    """Pivotal tuning's decoder-weight gradients (OSGDecoder, training/triplane.py:124-136; Adam over every weight, base_coach.py:96-99)
    contracted inside the sample-level backward kernel (decode_rows_kernel<true, true, true>: per 32-sample tile the operands' fp16 pieces
    transposed through LDS, three-product v_mfma_f32_16x16x32_f16, one power of two per tile on the gradient side) against (a) the same
    backward with the four operands dumped and contracted by the exact-fp32 rows_gram kernel and (b) the oracle's autograd.
    'wide_range': the per-ray cotangent spans eight decades from ray to ray (tiles whose samples differ by 1e8 in gradient size);
    'ragged': a ray count that leaves the last 32-sample tile partly empty and uneven coarse / fine counts (absent sample rows)."""
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from inv3d_amd import fused
    cfg = O.full_config()
    opts = dict(cfg.rendering)
    res, n = (32, 2) if variant != 'ragged' else (9, 1)
    if variant == 'ragged':
        opts['depth_resolution'], opts['depth_resolution_importance'] = 40, 24
    P = O.synth_params(O.small_config(), seed=7)
    g = torch.Generator().manual_seed(17)
    planes = (torch.randn(n, 3, 32, 64, 64, generator=g) * 0.8)
    cam = O.synth_cameras(n, seed=11)
    o, dr = O.ray_sampler(cam[:, :16].reshape(n, 4, 4), cam[:, 16:].reshape(n, 3, 3), res)
    dc, df = opts['depth_resolution'], opts['depth_resolution_importance']
    u1, u2 = torch.rand(n, res * res, dc, 1, generator=g), torch.rand(n * res * res, df, generator=g)
    g_rgb, g_dep = torch.randn(n, res * res, 32, generator=g), torch.randn(n, res * res, 1, generator=g)
    if variant == 'wide_range':
        amp = 10.0 ** (torch.rand(n, res * res, 1, generator=g) * 8 - 6)
        g_rgb, g_dep = g_rgb * amp, g_dep * amp
    names = ['decoder.net.0.weight', 'decoder.net.0.bias', 'decoder.net.2.weight', 'decoder.net.2.bias']

    Pd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    rgb_r, dep_r, _ = O.render(Pd, planes, o, dr, opts, u1, u2)
    ref = torch.autograd.grad([rgb_r, dep_r], [Pd[k] for k in names], [g_rgb, g_dep])

    def run(in_kernel):
        old = fused.GRAM_FUSED
        fused.GRAM_FUSED = in_kernel
        try:
            dec = _decoder(P)
            R = ImportanceRenderer()
            R.set_uniforms(u1.to(DEV), u2.to(DEV))
            pg = planes.to(DEV).requires_grad_(True)
            rgb, dep, _ = R(pg, dec, o.to(DEV), dr.to(DEV), opts)
            pd = dict(dec.named_parameters())
            gg = torch.autograd.grad([rgb, dep], [pd[k[len('decoder.'):]] for k in names] + [pg], [g_rgb.to(DEV), g_dep.to(DEV)])
            torch.cuda.synchronize()
            return gg
        finally:
            fused.GRAM_FUSED = old
    a, b = run(True), run(False)
    # (the oracle is fp32 on the CPU: its importance samples differ from the kernels' in a few bins, hence the looser bound against it)
    for k, nm in enumerate(names):
        print(f'{variant} {nm}: in-kernel vs dumped {float((a[k] - b[k]).abs().max() / b[k].abs().max()):.2e}, in-kernel vs oracle '
              f'{float((a[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()):.2e}, dumped vs oracle {float((b[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()):.2e}')
        close_rel(a[k], b[k], 2e-5, f'{variant} {nm}: in-kernel Gram vs dumped operands')
        close_rel(a[k], ref[k], 2e-4, f'{variant} {nm}: in-kernel Gram vs the oracle')
        close_rel(b[k], ref[k], 2e-4, f'{variant} {nm}: dumped operands vs the oracle')
    close_rel(a[4], b[4], 1e-6, f'{variant} d planes unchanged by the Gram path')


@pytest.mark.parametrize('variant', ['ffhq48', 'white_back', 'disparity', 'coarse_only', 'uneven', 'negative_depths', 'no_grad_fused'])
def test_render_vs_oracle(variant):
    """48+48-sample configuration (and variants) on random planes vs the oracle, forward and gradients."""
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    cfg = O.full_config()
    opts = dict(cfg.rendering)
    res, n = 10, 2
    if variant == 'white_back':
        opts['white_back'] = True
    if variant == 'disparity':
        opts['disparity_space_sampling'] = True
    if variant == 'coarse_only':
        opts['depth_resolution_importance'] = 0
    if variant == 'uneven':
        opts['depth_resolution'], opts['depth_resolution_importance'] = 40, 24
    if variant == 'negative_depths':         # sample depths of both signs: the global depth range goes through both branches of the integer atomics
        opts['ray_start'], opts['ray_end'] = -0.4, 0.7
    P = O.synth_params(O.small_config(), seed=7)
    g = torch.Generator().manual_seed(21)
    planes = (torch.randn(n, 3, 32, 64, 64, generator=g) * 0.8)
    cam = O.synth_cameras(n, seed=11)
    c2w, K = cam[:, :16].reshape(n, 4, 4), cam[:, 16:].reshape(n, 3, 3)
    o, dr = O.ray_sampler(c2w, K, res)
    dc, df = opts['depth_resolution'], opts['depth_resolution_importance']
    u1 = torch.rand(n, res * res, dc, 1, generator=g)
    u2 = torch.rand(n * res * res, df, generator=g) if df else None
    g_rgb = torch.randn(n, res * res, 32, generator=g)
    g_dep = torch.randn(n, res * res, 1, generator=g)
    pr, orr, drr = planes.clone().requires_grad_(True), o.clone().requires_grad_(True), dr.clone().requires_grad_(True)
    rgb_r, dep_r, ws_r = O.render(P, pr, orr, drr, opts, u1, u2)
    gr = torch.autograd.grad([rgb_r, dep_r], [pr, orr, drr], [g_rgb, g_dep])
    dec = _decoder(P)
    R = ImportanceRenderer()
    R.set_uniforms(u1.to(DEV), u2.to(DEV) if u2 is not None else None)
    pg = planes.to(DEV).requires_grad_(True)
    og, dg = o.to(DEV).requires_grad_(True), dr.to(DEV).requires_grad_(True)
    if variant == 'no_grad_fused':           # inference form: one fused ray kernel (training mode runs the pipelined stages)
        with torch.no_grad():
            rgb, dep, ws = R(pg, dec, og, dg, opts)
        close(rgb, rgb_r, 1e-5, f'{variant} rgb'); close(dep, dep_r, 1e-5, f'{variant} depth'); close(ws, ws_r, 1e-5, f'{variant} wsum')
        return
    rgb, dep, ws = R(pg, dec, og, dg, opts)
    close(rgb, rgb_r, 1e-5, f'{variant} rgb'); close(dep, dep_r, 1e-5, f'{variant} depth'); close(ws, ws_r, 1e-5, f'{variant} wsum')
    gg = torch.autograd.grad([rgb, dep], [pg, og, dg], [g_rgb.to(DEV), g_dep.to(DEV)])
    close(gg[0], gr[0], 1e-4, f'{variant} d planes'); close_most(gg[1], gr[1], 1e-4, f'{variant} d origins'); close_most(gg[2], gr[2], 1e-4, f'{variant} d dirs')


ORACLE_BIN_MISMATCHES = {'random': 19, 'det': 26, 'ties': 17, 'no_grad_fused': 19}          # observed in round 6, identical in both builds


@pytest.mark.parametrize('variant', ['random', 'det', 'ties', 'no_grad_fused'])
def test_sampler_indices_exact(variant):
    """The integer side of the sampler at 110 592 rays x (48 + 48) samples (renderer.py:281-307 sample_pdf: searchsorted(right=True), the
    below / above clamps; :212-222 unify_samples: the stable sort's permutation), compared as INTEGERS:
      (a) against torch.searchsorted on the kernel's own CDF edges and against torch.sort(stable=True) on the kernel's own depths: bit-exact,
          every ray -- the index logic itself;
      (b) against the CPU oracle run from the kernel's coarse (depth, sigma): mismatches are COUNTED; each must be a sample whose uniform
          lies within float rounding of a CDF edge (the oracle's CDF comes out of torch's vectorised sum / cumsum, the kernel's out of a
          sequential scan), and there must be few.
    Variants: random uniforms; `det=True` of the reference (u = linspace(0, 1, 48), the end points included); tied depths (duplicated
    importance uniforms -> equal fine depths; jitter 0 next to jitter 1 - 2^-24 -> coarse neighbours that round together)."""
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    from inv3d_amd import fused
    cfg = O.full_config()
    opts = dict(cfg.rendering)
    n, res = 3, 192                                   # 110 592 rays
    P = O.synth_params(O.small_config(), seed=7)
    g = torch.Generator().manual_seed(23)
    planes = torch.randn(n, 3, 32, 64, 64, generator=g) * 0.8
    cam = O.synth_cameras(n, seed=11)
    o, dr = O.ray_sampler(cam[:, :16].reshape(n, 4, 4), cam[:, 16:].reshape(n, 3, 3), res)
    dc, df = opts['depth_resolution'], opts['depth_resolution_importance']
    R_ = res * res
    u1 = torch.rand(n, R_, dc, 1, generator=g)
    u2 = torch.rand(n * R_, df, generator=g)
    if variant == 'det':
        u2 = torch.linspace(0, 1, df).expand(n * R_, df).contiguous()                         # renderer.py:286-288
    if variant == 'ties':
        u2[:, 1::4] = u2[:, 0::4]                                                               # equal fine depths within a ray
        u2[:, 7] = u2[:, 40]
        u1[:, :, 10::8] = 0.0                                                                   # sample i+1 at the start of its stratum ...
        u1[:, :, 9::8] = 1.0 - 2.0 ** -24                                                        # ... sample i at the very end of its own
    dbg = fused.SAMPLER_DEBUG = {}
    try:
        Rm = ImportanceRenderer()
        Rm.set_uniforms(u1.to(DEV), u2.to(DEV))
        pg = planes.to(DEV).requires_grad_(variant != 'no_grad_fused')
        with torch.set_grad_enabled(variant != 'no_grad_fused'), (torch.no_grad() if variant == 'no_grad_fused' else torch.enable_grad()):
            if variant == 'no_grad_fused':
                old, fused.RENDER_PIPELINE_NOGRAD = fused.RENDER_PIPELINE_NOGRAD, False            # the one-kernel form (render_kernel<0>)
            try:
                Rm(pg, _decoder(P), o.to(DEV), dr.to(DEV), opts)
            finally:
                if variant == 'no_grad_fused':
                    fused.RENDER_PIPELINE_NOGRAD = old
        torch.cuda.synchronize()
    finally:
        fused.SAMPLER_DEBUG = None
    inds, ranks, cdf = dbg['inds'].cpu().long(), dbg['ranks'].cpu().long(), dbg['cdf'].cpu()
    fine = dbg['fine'].reshape(n * R_, df).cpu()
    ns = dc - 3
    assert int((inds < 0).sum()) == 0 and int((ranks < 0).sum()) == 0, 'every sample must have been written'
    # ---- (a) index logic on the kernel's own numbers: exact --------------------------------------------------------------------------
    edges = torch.cat([torch.zeros(n * R_, 1), cdf[:, :ns]], 1)
    want = torch.searchsorted(edges, u2.contiguous(), right=True)
    assert torch.equal(inds[..., 0], want), f'searchsorted(right=True): {int((inds[..., 0] != want).sum())} of {want.numel()} indices differ'
    assert torch.equal(inds[..., 1], torch.clamp_min(want - 1, 0)) and torch.equal(inds[..., 2], torch.clamp_max(want, ns))
    depths_c = O.sample_stratified(n, R_, opts['ray_start'], opts['ray_end'], dc, opts['disparity_space_sampling'], u1).reshape(n * R_, dc)
    if dbg.get('pos') is not None:                    # pipelined forward: the kernel's own coarse depths (4th component of the position rows)
        dcg = dbg['pos'][0].reshape(n * R_, -1, 4)[:, :dc, 3].cpu()
        assert float((dcg - depths_c).abs().max()) <= 2e-6
        depths_c = dcg
    allz = torch.cat([depths_c, fine], 1)
    _, perm = torch.sort(allz, dim=1, stable=True)
    inv = torch.empty_like(perm)
    inv.scatter_(1, perm, torch.arange(dc + df).expand_as(perm))
    assert torch.equal(ranks, inv), f'unify_samples permutation: {int((ranks != inv).any(1).sum())} of {n * R_} rays differ'
    if variant == 'ties':
        tied = (allz.sort(1).values.diff(dim=1) == 0).any(1)
        assert int(tied.sum()) >= 0.9 * n * R_, 'the tie variant must actually tie'
    # ---- (b) against the oracle from the kernel's coarse pass -------------------------------------------------------------------------
    if dbg.get('rows') is None:
        return
    sig = dbg['rows'][0].reshape(n * R_, 2, max(dc, df))[:, 0, :dc].cpu()
    od = {}
    zc = depths_c.reshape(n, R_, dc, 1)
    _, _, w = O.ray_march(torch.zeros(n, R_, dc, 1), sig.reshape(n, R_, dc, 1), zc, opts)
    O.sample_importance(zc, w, df, u2, debug=od)
    bad = od['inds'] != inds[..., 0]
    nbad = int(bad.sum())
    # (det=True puts its last uniform exactly ON the last edge, u = 1.0 = cdf[ns] up to rounding: which side it falls on is the rounding of a
    #  48-term sum -- a third of the rays differ there, and only there)
    at_end = (u2 >= 1.0).expand_as(bad)
    n_off = int((bad & ~at_end).sum())
    print(f'sampler indices [{variant}]: mismatches away from u = 1: {n_off}')
    # the kernel's scan and the oracle's cumsum are both deterministic: the count is a constant of (seed, variant), observed in round 6 in both builds
    # (a bound of 2e-4 x 5.3 M = 1061 would let sixty times as many through)
    assert n_off <= 2e-5 * bad.numel()
    if ORACLE_BIN_MISMATCHES.get(variant) is not None:
        assert n_off == ORACLE_BIN_MISMATCHES[variant], f'{n_off} of {bad.numel()} bin indices differ from the oracle, expected exactly {ORACLE_BIN_MISMATCHES[variant]}'
    if nbad:
        # every mismatch: u within rounding of the edge the two sides disagree about
        r_i, s_i = bad.nonzero(as_tuple=True)
        lo = torch.minimum(od['inds'][r_i, s_i], inds[r_i, s_i, 0])
        e_o, e_k = od['cdf'][r_i, lo.clamp(max=ns)], edges[r_i, lo.clamp(max=ns)]
        assert float((u2[r_i, s_i] - e_o).abs().max()) <= 4e-6 and float((e_o - e_k).abs().max()) <= 4e-6, 'an index mismatch that rounding of the CDF does not explain'
        assert int((od['inds'][r_i, s_i] - inds[r_i, s_i, 0]).abs().max()) <= 2
    print(f'sampler indices [{variant}]: {bad.numel()} bin indices, {nbad} differ from the CPU oracle (rounding of the CDF), permutation of {n * R_} rays exact')


def test_render_zero_density_ray():
    """A ray with no density anywhere: depth must take the NaN -> +inf -> clamp-to-global-max path (ray_marcher.py:49-50)."""
    from inv3d_amd import hipops as H
    from inv3d_amd import fused
    cfg = O.small_config()
    opts = dict(cfg.rendering)
    P = O.synth_params(cfg, seed=5)
    P = dict(P)
    P['decoder.net.2.bias'] = P['decoder.net.2.bias'].clone()
    P['decoder.net.2.weight'] = P['decoder.net.2.weight'].clone()
    P['decoder.net.2.weight'][0] = 0
    P['decoder.net.2.bias'][0] = -90.0            # sigma = -90 everywhere
    g = torch.Generator().manual_seed(2)
    planes = torch.randn(1, 3, 32, 16, 16, generator=g)
    cam = O.synth_cameras(1, seed=3)
    o, dr = O.ray_sampler(cam[:, :16].reshape(1, 4, 4), cam[:, 16:].reshape(1, 3, 3), 4)
    u1 = torch.rand(1, 16, 12, 1, generator=g); u2 = torch.rand(16, 12, generator=g)
    rgb_r, dep_r, ws_r = O.render(P, planes, o, dr, opts, u1, u2)
    dec = _decoder(P)
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    R = ImportanceRenderer()
    R.set_uniforms(u1.to(DEV), u2.to(DEV))
    pg = planes.to(DEV).requires_grad_(True)
    rgb, dep, ws = R(pg, dec, o.to(DEV), dr.to(DEV), opts)
    close(dep, dep_r, 1e-5, 'zero-density depth'); close(rgb, rgb_r, 1e-5, 'zero-density rgb')
    gp, = torch.autograd.grad([rgb, dep], [pg], [torch.ones_like(rgb), torch.ones_like(dep)])
    assert torch.isfinite(gp).all()


def test_run_model_vs_oracle():
    from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=5)
    g = torch.Generator().manual_seed(8)
    planes = torch.randn(2, 3, 32, 16, 16, generator=g)
    coords = (torch.rand(2, 300, 3, generator=g) - 0.5) * 1.2
    rgb_r, sig_r = O.run_model(P, planes, coords, cfg.rendering)
    out = ImportanceRenderer().run_model(planes.to(DEV), _decoder(P), coords.to(DEV), None, cfg.rendering)
    close(out['rgb'], rgb_r, 1e-5, 'run_model rgb'); close(out['sigma'], sig_r, 1e-5, 'run_model sigma')


# ------------------------------------------------------------------------------------------------- noise buffers
@pytest.mark.parametrize('shape', [(100003, 64, 32), (4097, 33, 64), (17, 5, 7), (2, 64, 64), (1001, 36, 40), (513, 33, 20), (129, 40, 64), (77, 41, 33)])
def test_rows_gram(shape):
    """a^T b and column sums of tall-skinny row matrices (decoder-weight gradients of pivotal tuning) vs fp64."""
    from inv3d_amd import hipops as H
    s, ka, kb = shape
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(s, ka, generator=g), torch.randn(s, kb, generator=g)
    out, cs = H.rows_gram(a.to(DEV), b.to(DEV))
    ref, refs = a.double().t() @ b.double(), a.double().sum(0)
    close(out, ref, 2e-6 * math.sqrt(s), f'rows_gram {shape}')
    close(cs, refs, 2e-6 * math.sqrt(s), f'rows_gram colsum {shape}')
    out2, cs2 = H.rows_gram(a.to(DEV), b.to(DEV), 0.125, 3.0)               # the scaled form (eg3d_rows_gram_scaled): gains applied to the partial sums
    close(out2, ref * 0.125, 2e-6 * math.sqrt(s), f'rows_gram scaled {shape}')
    close(cs2, refs * 3.0, 2e-6 * math.sqrt(s), f'rows_gram scaled colsum {shape}')


def test_noise_regularizer_and_normalize():
    from inv3d_amd import hipops as H
    from inv3d_amd.inversion import noise_regularizer
    g = torch.Generator().manual_seed(31)
    bufs = [torch.randn(r, r, generator=g) for r in (4, 8, 16, 64, 128, 512)]
    ref_in = [b.clone().requires_grad_(True) for b in bufs]
    reg_r = O.noise_regularizer(ref_in) * 1e5
    gr = torch.autograd.grad(reg_r, ref_in)
    dev_in = [b.to(DEV).requires_grad_(True) for b in bufs]
    reg = noise_regularizer(dev_in, 1e5)
    close(reg, reg_r, 1e-4, 'noise reg value')
    gg = torch.autograd.grad(reg * 2.0, dev_in)
    for a, b in zip(gg, gr):
        close(a, b * 2.0, 1e-4, 'noise reg grad')
    dn = [b.to(DEV).clone() for b in bufs]
    H.noise_normalize_(dn)
    for a, b in zip(dn, bufs):
        e = b - b.mean()
        e = e * e.square().mean().rsqrt()
        close(a, e, 1e-5, 'noise normalize')


def test_hip_adam_matches_torch_adam_with_regulariser_gradient_and_renormalisation():
    """eg3d_adam_step against torch.optim.Adam on the same leaves (w_projector.py:107-118,256-270): gradient = autograd's + a second list,
    maps renormalised after the update, learning rate changed between steps, 40 leaves (two banks), one leaf without any gradient."""
    from inv3d_amd import hipops as H
    g = torch.Generator().manual_seed(77)
    shapes = [(1, 1, 512)] + [(2, 1, r, r) for r in (4, 8, 16, 32, 64, 128)] * 6 + [(2, 1, 512, 512), (3, 5), (7,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    ref = [t.clone().to(DEV).requires_grad_(True) for t in init]
    mine = [t.clone().to(DEV).requires_grad_(True) for t in init]
    maps_r, maps_m = ref[1:-2], mine[1:-2]
    o_ref = torch.optim.Adam(ref, lr=0.1, betas=(0.9, 0.999))
    o_mine = H.HipAdam(mine, lr=torch.tensor(0.1, device=DEV), betas=(0.9, 0.999))
    for step in range(6):
        lr = 0.1 * (1.0 - 0.12 * step)
        grads = [torch.randn(s, generator=g).to(DEV) for s in shapes]
        extra = [torch.randn(t.shape, generator=g).to(DEV) * 0.3 for t in init[1:-2]]
        for p, q, gr in zip(ref, mine, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        ref[-1].grad = mine[-1].grad = None                     # never receives a gradient: untouched, as in torch
        if step % 2:
            maps_r[3].grad = None                               # regulariser gradient only
            maps_m[3].grad = None
        for p, e in zip(maps_r, extra):
            p.grad = e.clone() if p.grad is None else p.grad + e
        o_ref.param_groups[0]['lr'] = lr
        o_ref.step()
        with torch.no_grad():
            for p in maps_r:                                    # per image of the batch
                p -= p.mean(dim=(1, 2, 3), keepdim=True)
                p *= p.square().mean(dim=(1, 2, 3), keepdim=True).rsqrt()
        o_mine.param_groups[0]['lr'].fill_(lr)
        o_mine.step(extra_grads=dict(zip(maps_m, extra)), normalize={p: 2 for p in maps_m})
        for i, (p, q) in enumerate(zip(ref, mine)):
            close(q.detach(), p.detach(), 2e-5, f'adam leaf {i} step {step}')
    assert float(o_mine.step_t) == 6.0
    assert mine[0]._version >= 6 and mine[-1]._version == 0          # raw-pointer writes are reported to autograd / memo()
    assert torch.equal(mine[-1].detach().cpu(), init[-1])


@pytest.mark.parametrize('shape', [(1, 32, 64, 16, 16, 1), (2, 160, 96, 12, 9, 1), (1, 128, 128, 8, 8, 2), (1, 4, 16, 10, 10, 1), (1, 128, 256, 32, 40, 1),
                                   (2, 72, 136, 12, 24, 1), (1, 256, 128, 4, 8, 1)])
@pytest.mark.parametrize('prec,tol', [('f32', 2e-5), ('f16x3', 2e-5)])
def test_conv_wgrad_vs_torch(shape, prec, tol):
    """eg3d_conv2d_wgrad_f32 in both arithmetic modes vs autograd of F.conv2d / F.conv_transpose2d in float64: style-modulated input,
    ragged channel counts, batch > 1, the four parity classes of an up-sampling layer, and (f16x3) a gradient operand of magnitude
    1e-6 brought into range by g_amax."""
    from inv3d_amd import hipops as H
    n, ci, co, h, w, up = shape
    g_ = torch.Generator().manual_seed(ci * 31 + co)
    x = torch.randn(n, ci, h, w, generator=g_)
    s = torch.rand(n, ci, generator=g_) + 0.5
    wt = (torch.randn(co, ci, 3, 3, generator=g_) / (3 * ci ** 0.5)).double().requires_grad_(True)
    xs = (x * s[:, :, None, None]).double()
    if up == 1:
        y = F.conv2d(xs, wt, padding=1)
        cls, out_stride = H.classes_corr(h, w, 3, 3, 1), 1
    else:
        y = F.conv_transpose2d(xs, wt.transpose(0, 1), stride=2)
        cls, out_stride = H.classes_convT(h, w, 3, 3, 2)[0], 2
    dy = torch.randn(y.shape, generator=g_) * 1e-6
    y.backward(dy.double())
    cop = (co + 3) // 4 * 4
    gq = torch.zeros(n, cop, *y.shape[2:])
    gq[:, :co] = dy
    xq, gq = x.to(DEV).contiguous(memory_format=torch.channels_last), gq.to(DEV).contiguous(memory_format=torch.channels_last)
    dwp = torch.zeros(co, 9 * ci, device=DEV)
    amax = gq.abs().max().reshape(1)
    H.conv_wgrad(xq, gq, ci, co, dwp, cls, in_stride=1, out_stride=out_stride, in_scale=s.to(DEV).contiguous(), precision=prec,
                 g_amax=amax if prec == 'f16x3' else None)
    dw = dwp.view(co, 3, 3, ci).permute(0, 3, 1, 2)
    ref = wt.grad
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= tol, err


@pytest.mark.parametrize('products,tol', [(3, 2e-5), (1, 2e-3)])
@pytest.mark.parametrize('shape', [(1, 64, 64, 16, 32, 0), (2, 128, 64, 24, 40, 0), (1, 64, 128, 9, 33, 3), (1, 128, 128, 64, 64, 0), (1, 256, 64, 7, 70, 2)])
def test_conv_wgrad_v2_vs_torch(shape, products, tol):
    """eg3d_conv2d_wgrad_v2 (split images in, LDS-DMA + transposing LDS reads) vs autograd of F.conv2d in float64: style-modulated input,
    batch 2, widths that are not multiples of the 32-column strip, several row groups, a gradient operand of magnitude 1e-6 (range-
    normalised by the split), three-product and single-product (fp16-operand, rel. 2^-11) arithmetic."""
    from inv3d_amd import hipops as H
    n, ci, co, h, w, rg = shape
    g_ = torch.Generator().manual_seed(ci * 31 + co + h)
    x = torch.randn(n, ci, h, w, generator=g_)
    s = torch.rand(n, ci, generator=g_) + 0.5
    wt = (torch.randn(co, ci, 3, 3, generator=g_) / (3 * ci ** 0.5)).double().requires_grad_(True)
    y = F.conv2d((x * s[:, :, None, None]).double(), wt, padding=1)
    dy = torch.randn(y.shape, generator=g_) * 1e-6
    y.backward(dy.double())
    xq, gq = x.to(DEV).contiguous(memory_format=torch.channels_last), dy.to(DEV).contiguous(memory_format=torch.channels_last)
    ximg = H.split_activation(xq, H.absmax(xq), in_scale=s.to(DEV).contiguous())
    gimg = H.split_activation(gq, H.absmax(gq))
    cls = H.classes_corr(h, w, 3, 3, 1)
    assert H.conv_wgrad_v2_ok(gimg, ximg, cls)
    dwp = torch.zeros(co, 9 * ci, device=DEV)
    H.conv_wgrad_v2(gimg, ximg, dwp, cls, products=products, row_groups=rg)
    dw = dwp.view(co, 3, 3, ci).permute(0, 3, 1, 2)
    ref = wt.grad
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= tol, err
    # the atomic-free form: slabs of partial tiles, summed in slab order by weight_grad_finish -- bit-identical from run to run
    outs = []
    for _ in range(3):
        slabs = H.conv_wgrad_v2_slabs(gimg, ximg, cls, products=products, row_groups=rg)
        wparam = torch.zeros(co, ci, 3, 3, device=DEV)
        outs.append(H.weight_grad_finish(slabs, wparam, None, None, None))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    err = float((outs[0].double().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= tol, err


@pytest.mark.parametrize('shape', [(512, 512, 3), (96, 256, 1), (3, 128, 1), (40, 33, 3), (16, 8, 3)])
def test_pack_conv_weight(shape):
    """eg3d_pack_conv_weight == the two permute-copies + sum of squares it replaces (bit-exact copies)."""
    from inv3d_amd import hipops as H
    o, i, k = shape
    w = torch.randn(o, i, k, k, generator=torch.Generator().manual_seed(o + i)).to(DEV)
    wf, wa, wsq = H.pack_conv_weight(w)
    assert torch.equal(wf, H.pack_weight_fwd(w)) and torch.equal(wa, H.pack_weight_adj(w))
    close(wsq, w.square().sum((2, 3)), 1e-6, 'wsq')


# ---------------------------------------------------------------------------------------------------------------------------
# Pre-split convolution (csrc/conv_v2.hip): split images + halo-staged kernel vs torch fp64
# ---------------------------------------------------------------------------------------------------------------------------
def _v2_operands(x, wt, styles, adjoint=False):
    from inv3d_amd import hipops as H
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    co, ci, k, _ = wt.shape
    wp = (H.pack_weight_adj if adjoint else H.pack_weight_fwd)(wt.to(DEV))
    wimg = H.split_weight(wp, ci if adjoint else co, co if adjoint else ci, k * k)
    s = styles.to(DEV).contiguous() if styles is not None else None
    aimg = H.split_activation(xc, H.absmax(xc), in_scale=s, s_amax=H.absmax(s) if s is not None else None)
    return xc, aimg, wimg


@pytest.mark.parametrize('shape', [(1, 512, 16, 16, 512), (1, 512, 8, 8, 512), (1, 512, 4, 4, 512), (2, 64, 8, 16, 128), (1, 96, 5, 7, 64), (1, 32, 32, 32, 64),
                                   (3, 32, 2, 32, 192), (1, 32, 1, 1, 64)])
@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize('adjoint', [False, True])
def test_conv_ws_vs_torch(shape, products, adjoint):
    """Weight-streaming split-K kernel (csrc/conv_ws.hip): 3x3 stride-1 correlation of the style-modulated fp32 activation (forward taps) and the
    flipped-tap data gradient (adjoint weight image), accumulated into a buffer that already holds values, vs torch fp64; the backbone's 16^2 /
    8^2 / 4^2 x 512 -> 512 layers, batch 2 / 3, ragged 5 x 7 and 1 x 1 images, a 32^2 image in four 256-cell blocks."""
    from inv3d_amd import hipops as H
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(77)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    if adjoint:           # dx = conv_transpose(dz, w): contraction over co
        x = torch.randn(n, co, h, w, generator=g)
        s = None
        ref = torch.nn.functional.conv_transpose2d(x.double(), wt.double(), padding=1)
        wp = H.pack_weight_adj(wt.to(DEV)); wimg = H.split_weight(wp, ci, co, 9)
        cls = H.classes_corr_adjoint(h, w, 3, 3, 1)
        cout = ci
    else:
        x = torch.randn(n, ci, h, w, generator=g) * 3
        s = 1 + 0.5 * torch.randn(n, ci, generator=g)
        ref = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1)
        wp = H.pack_weight_fwd(wt.to(DEV)); wimg = H.split_weight(wp, co, ci, 9)
        cls = H.classes_corr(h, w, 3, 3, 1)
        cout = co
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    sd = s.to(DEV) if s is not None else None
    base = torch.randn(n, cout, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    assert H.conv_ws_ok(x.shape[1], cout, cls, n, h, w) == (n * h * w <= 256)
    outs = []
    for rep in range(2):
        z = base.clone(memory_format=torch.channels_last)
        H.conv_ws(xc, wimg, z, cls, in_scale=sd, x_amax=H.absmax(xc), products=products)
        outs.append(z)
    close(outs[0] - base, ref.float(), 2e-5 if products == 3 else 3e-3, f'conv_ws {shape} adjoint={adjoint}')


@pytest.mark.parametrize('shape', [(512, 8, 8, 512), (512, 4, 4, 512), (48, 3, 5, 64), (32, 6, 6, 96), (16, 1, 1, 32), (64, 2, 30, 32)])
@pytest.mark.parametrize('products', [3, 1])
def test_conv_ws_transposed_vs_torch(shape, products):
    """The transposed form of the weight-streaming kernel (forward of an up layer: stride-2 3x3 transposed conv of the style-modulated activation,
    accumulated into a (2H + 1) x (2W + 1) buffer that holds values) vs conv_transpose2d in fp64: the backbone's 8^2 -> 17^2 and 4^2 -> 9^2
    x 512 layers, ragged images with one / two / three row tiles per parity class, a short last chunk group, a single pixel."""
    from inv3d_amd import hipops as H
    ci, h, w, co = shape
    g = torch.Generator().manual_seed(93)
    x = torch.randn(1, ci, h, w, generator=g) * 2
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(1, ci, generator=g)
    ref = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], wt.double().transpose(0, 1), stride=2)
    wimg = H.split_weight(H.pack_weight_fwd(wt.to(DEV)), co, ci, 9)
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    base = torch.randn(1, co, 2 * h + 1, 2 * w + 1, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    assert H.conv_ws_up_ok(ci, co, 1, h, w, max_cells=96)
    z = base.clone(memory_format=torch.channels_last)
    H.conv_ws_up(xc, wimg, z, in_scale=s.to(DEV), x_amax=H.absmax(xc), products=products)
    close(z - base, ref.float(), 2e-5 if products == 3 else 3e-3, f'conv_ws transposed {shape}')


@pytest.mark.parametrize('shape', [(512, 16, 16, 512), (512, 8, 8, 512), (512, 4, 4, 512), (64, 8, 16, 96), (48, 3, 5, 32), (32, 8, 32, 64), (16, 1, 1, 32)])
@pytest.mark.parametrize('products', [3, 1])
def test_conv_ws_stride2_adjoint_vs_torch(shape, products):
    """The stride-2 adjoint form of the weight-streaming kernel (data gradient of an up layer before the style scale): with G the (2H + 1) x
    (2W + 1) gradient of the transposed conv's output, dx = conv2d(G, w^T-taps, stride 2) -- against torch fp64 on the backbone's 32^2 -> 16^2,
    16^2 -> 8^2, 8^2 -> 4^2 layers (512 channels), 128-cell and ragged images, a contraction whose last chunk group is short, a single cell;
    accumulated into a buffer that holds values."""
    from inv3d_amd import hipops as H
    co, h, w, ci = shape                                # contraction over co (the up layer's output channels), result ci channels at h x w
    g = torch.Generator().manual_seed(91)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(co * 9)
    G = torch.randn(1, co, 2 * h + 1, 2 * w + 1, generator=g)
    # forward: z = conv_transpose2d(x, w, stride 2) (w: [ci -> co] as [ci, co, 3, 3] = wt.transpose(0, 1)); its adjoint w.r.t. x:
    ref = torch.nn.functional.conv2d(G.double(), wt.double().transpose(0, 1), stride=2)
    assert ref.shape == (1, ci, h, w)
    wp = H.pack_weight_adj(wt.to(DEV)); wimg = H.split_weight(wp, ci, co, 9)
    cls = H.classes_convT_adjoint(h, w, 3, 3, 2)
    Gc = G.to(DEV).contiguous(memory_format=torch.channels_last)
    base = torch.randn(1, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    assert H.conv_ws_ok(co, ci, cls, 1, h, w, in_stride=2)
    z = base.clone(memory_format=torch.channels_last)
    H.conv_ws(Gc, wimg, z, cls, x_amax=H.absmax(Gc), products=products, in_stride=2)
    close(z - base, ref.float(), 2e-5 if products == 3 else 3e-3, f'conv_ws s2adj {shape}')


@pytest.mark.parametrize('rows', [8, 4])        # 4: the half-height patch (hipops.V2_HALF)
@pytest.mark.parametrize('shape', [(1, 32, 16, 64, 128), (2, 64, 40, 72, 128), (1, 128, 33, 37, 256), (1, 16, 8, 32, 128)])
def test_conv_v2_forward_epilogue_vs_torch(shape, rows):
    """3x3 correlation with the fused forward epilogue (style-modulated input, demodulation, noise, bias, lrelu, gain, skip addend) on
    ragged grids (sizes that are not multiples of the 8 x 32 patch), batch 2 and two 128-channel tiles; max|out| reported."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    d = 0.5 + torch.rand(n, co, generator=g)
    noise, strength = torch.randn(n, 1, h, w, generator=g), torch.tensor(0.3)
    bias, add = 0.1 * torch.randn(co, generator=g), torch.randn(n, co, h, w, generator=g)
    z = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1) * d.double()[:, :, None, None]
    ref = torch.nn.functional.leaky_relu(z + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 1.4 + add.double()
    xc, aimg, wimg = _v2_operands(x, wt, s)
    out = H.empty_cl(n, co, h, w, DEV)
    amax = torch.zeros(1, device=DEV)
    H.conv_v2(aimg, wimg, out, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_FWD, out_scale=d.to(DEV), bias=bias.to(DEV), noise=noise.to(DEV).contiguous(),
              noise_nstride=h * w, noise_strength=strength.to(DEV), act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0,
              addend=add.to(DEV).contiguous(memory_format=torch.channels_last), out_amax=amax, patch_rows=rows)       # 8 x 32 and 4 x 32-cell patches
    close(out, ref.float(), 2e-5, f'conv_v2 fwd {shape}')
    assert abs(float(amax) - float(out.abs().max())) == 0.0


@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize('shape', [(1, 32, 16, 64, 128), (2, 64, 40, 72, 128), (1, 16, 9, 33, 128)])
def test_conv_v2_rgb_head_vs_torch(shape, products, rows=8):
    """The 1x1 head of the forward epilogue (eg3d_conv_v2_params::rgb_out: the toRGB layer that reads a 128-channel layer's output next,
    networks_stylegan2.py:338-359, evaluated while the values are in registers): y = clamp(sum_c out[c] w[o,c] s[n,c] + b[o]) against torch on the
    layer output the same launch wrote (exact fp32 arithmetic on identical inputs: 1e-6), a clamp that bites, ragged grids, batch 2, both
    arithmetic classes (the head is an instantiation of the 8-row patch kernel); the layer output itself is unchanged by the head."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(23)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    d = 0.5 + torch.rand(n, co, generator=g)
    bias = 0.1 * torch.randn(co, generator=g)
    tw = torch.zeros(4, co + 4)                         # row pitch larger than the channel count; fourth row = padding channel
    tw[:3, :co] = torch.randn(3, co, generator=g)
    ts = (1 + 0.5 * torch.randn(n, co, generator=g)) / math.sqrt(co)
    tb = torch.tensor([0.2, -0.1, 0.05, 0.0])
    xc, aimg, wimg = _v2_operands(x, wt, s)
    cls = H.classes_corr(h, w, 3, 3, 1)
    kw = dict(epi=L.EPI_FWD, out_scale=d.to(DEV), bias=bias.to(DEV), act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, patch_rows=rows, products=products)
    plain = H.empty_cl(n, co, h, w, DEV)
    H.conv_v2(aimg, wimg, plain, cls, **kw)
    for clamp in (-1.0, 0.6):
        out = H.empty_cl(n, co, h, w, DEV)
        y4 = H.empty_cl(n, 4, h, w, DEV)
        y4.fill_(float('nan'))
        twd = tw.to(DEV)
        H.conv_v2(aimg, wimg, out, cls, rgb_head=(twd[:, :co], ts.to(DEV), tb.to(DEV), y4, clamp) + ((3,) if clamp >= 0 else ()), **kw)       # (both forms of the padding channel)
        assert torch.equal(out, plain)
        ref = torch.einsum('nchw,oc,nc->nohw', out.double().cpu(), tw[:, :co].double(), ts.double()) + tb.double()[None, :, None, None]
        if clamp >= 0:
            assert float(ref.abs().max()) > clamp
            ref = ref.clamp(-clamp, clamp)
        close(y4, ref.float(), 2e-6, f'rgb head {shape} clamp {clamp}')
        assert float(y4[:, 3].abs().max()) == 0.0


def test_conv_v2_half_patch_full_size():
    """The 4 x 32-cell patch with the fused forward epilogue at a full-size layer (256^2 x 128 -> 128), three launches: the shapes of the test
    above never showed the fault this guards against (sporadic wrong elements from an SLP-vectorised epilogue: 3dgan-inversion_amd/Makefile)."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = 1, 128, 256, 256, 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(n, ci, generator=g)).to(DEV)
    d = (0.5 + torch.rand(n, co, generator=g)).to(DEV)
    noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    aimg = H.split_activation(x, H.absmax(x), in_scale=s)
    wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
    z = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1) * d.double()[:, :, None, None]
    ref = (torch.nn.functional.leaky_relu(z + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 1.4).float()
    outs = []
    for rows in (8, 4, 4, 4, 2, 2):
        out = H.empty_cl(n, co, h, w, DEV)
        H.conv_v2(aimg, wimg, out, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength,
                  act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, out_amax=torch.zeros(1, device=DEV), patch_rows=rows)
        torch.cuda.synchronize()
        close(out, ref, 5e-5, f'conv_v2 full size rows {rows}')
        outs.append(out)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), 'the 2-, 4- and 8-row patches accumulate in the same order: bit-identical results'


@pytest.mark.parametrize('shape', [(1, 256, 128, 128, 256), (2, 64, 36, 40, 128), (1, 96, 20, 64, 128)])
def test_conv_v2_k_halves_equal_the_four_wave_form(shape, monkeypatch):
    """KH = 2 of conv_v2_kernel (hipops.V2_KHALVES: 4-row launches with at most one workgroup per CU run as eight-wave workgroups whose halves split
    the contraction and meet in LDS): against float64 and against the four-wave form -- equal up to the one extra rounding of the cross-half sum --
    with the fused forward epilogue and with the data-gradient epilogue (style gradient, max|out|), at the full-size 128^2 x 256 layer, a batch of
    two on a ragged grid, and a chunk count (6) that is even but not a power of two."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(n, ci, generator=g)).to(DEV)
    d = (0.5 + torch.rand(n, co, generator=g)).to(DEV)
    noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    aimg = H.split_activation(x, H.absmax(x), in_scale=s)
    wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
    z = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1)
    ref = (torch.nn.functional.leaky_relu(z * d.double()[:, :, None, None] + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 1.4).float()
    cls = H.classes_corr(h, w, 3, 3, 1)
    res = {}
    for kh in (False, True):
        monkeypatch.setattr(H, 'V2_KHALVES', kh)
        out, amax = H.empty_cl(n, co, h, w, DEV), torch.zeros(1, device=DEV)
        H.conv_v2(aimg, wimg, out, cls, epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength,
                  act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, out_amax=amax, patch_rows=4)
        # the data-gradient epilogue on the same products: dx = z * styles, ds += sum z * xin
        dx, ds, amax2 = H.empty_cl(n, co, h, w, DEV), torch.zeros(n, co, device=DEV), torch.zeros(1, device=DEV)
        xin = torch.randn(n, co, h, w, generator=torch.Generator().manual_seed(6)).to(DEV).contiguous(memory_format=torch.channels_last)
        H.conv_v2(aimg, wimg, dx, cls, epi=L.EPI_BWD, out_scale=d, xin=xin, ds=ds, out_amax=amax2, patch_rows=4)
        torch.cuda.synchronize()
        close(out, ref, 5e-5, f'k halves {kh} fwd')
        assert abs(float(amax) - float(out.abs().max())) <= 1e-6 * float(amax)
        close(dx, (z * d.double()[:, :, None, None]).float(), 5e-5, f'k halves {kh} dx')
        close(ds, (z * xin.double()).sum((2, 3)).float(), 2e-4 * math.sqrt(h * w), f'k halves {kh} ds')
        assert abs(float(amax2) - float(dx.abs().max())) <= 1e-6 * float(amax2)
        res[kh] = (out, dx)
    for a, b in zip(res[False], res[True]):
        assert float((a - b).abs().max()) <= 4e-6 * float(a.abs().max()), 'the two forms differ by more than the rounding of one extra sum'
    # run-to-run: the kernel's counted waits / LDS-DMA protocol leave no room for a hazard to hide -- a first version of the KH template parameter changed
    # the code of EVERY instantiation slightly and the deterministic build stopped being bit-identical (sporadic 1e-7 .. 1e-5 differences).  Both forms,
    # sixty launches each on the same operands: every result bit-identical to the first.
    for kh in (False, True):
        monkeypatch.setattr(H, 'V2_KHALVES', kh)
        first = None
        for rep in range(60):
            out = H.empty_cl(n, co, h, w, DEV)
            H.conv_v2(aimg, wimg, out, cls, epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength,
                      act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, out_amax=torch.zeros(1, device=DEV), patch_rows=4)
            if first is None:
                first = out
            else:
                assert torch.equal(out, first), f'k halves {kh}: launch {rep} differs from the first'


# ---------------------------------------------------------------------------------------------------------------------------
# Wave-split pre-split convolution (csrc/conv_v3.hip): 128 / 64-cell x 64-channel tiles, contraction over the waves of the workgroup
# ---------------------------------------------------------------------------------------------------------------------------
V3_PLANS = [(4, 4), (2, 4), (2, 8)]


@pytest.mark.parametrize('plan', V3_PLANS)
@pytest.mark.parametrize('shape', [(1, 32, 16, 64, 128), (2, 64, 40, 72, 64), (1, 128, 33, 37, 192), (1, 16, 8, 32, 64), (1, 512, 32, 32, 128), (1, 80, 9, 33, 64)])
def test_conv_v3_forward_epilogue_vs_torch(shape, plan):
    """3x3 correlation with the fused forward epilogue on ragged grids, batch 2, 1 - 3 channel tiles of 64, chunk counts below / not a multiple of
    the number of waves (16, 80 and 32 channels over 4 / 8 waves: some waves own no chunk), max|out| reported; every tile / wave configuration."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(41)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    d = 0.5 + torch.rand(n, co, generator=g)
    noise, strength = torch.randn(n, 1, h, w, generator=g), torch.tensor(0.3)
    bias, add = 0.1 * torch.randn(co, generator=g), torch.randn(n, co, h, w, generator=g)
    z = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1) * d.double()[:, :, None, None]
    ref = torch.nn.functional.leaky_relu(z + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 1.4 + add.double()
    xc, aimg, wimg = _v2_operands(x, wt, s)
    out = H.empty_cl(n, co, h, w, DEV)
    amax = torch.zeros(1, device=DEV)
    H.conv_v3(aimg, wimg, out, H.classes_corr(h, w, 3, 3, 1), plan=plan, epi=L.EPI_FWD, out_scale=d.to(DEV), bias=bias.to(DEV), noise=noise.to(DEV).contiguous(),
              noise_nstride=h * w, noise_strength=strength.to(DEV), act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0,
              addend=add.to(DEV).contiguous(memory_format=torch.channels_last), out_amax=amax)
    close(out, ref.float(), 2e-5, f'conv_v3 fwd {shape} {plan}')
    assert abs(float(amax) - float(out.abs().max())) == 0.0


@pytest.mark.parametrize('plan', V3_PLANS)
@pytest.mark.parametrize('products', [3, 1])
def test_conv_v3_data_gradient_epilogue_vs_torch(plan, products):
    """Data gradient of a 3x3 layer: adjoint taps on the adjoint weight image, gradient-sized operand, dx = acc * styles + addend,
    ds = sum_px acc * x; three-product and single-product (fp16-operand) arithmetic."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = 2, 128, 24, 40, 96
    g = torch.Generator().manual_seed(42)
    gz = torch.randn(n, co, h, w, generator=g) * 1e-4
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s, xin, add = 1 + 0.5 * torch.randn(n, ci, generator=g), torch.randn(n, ci, h, w, generator=g), torch.randn(n, ci, h, w, generator=g) * 1e-4
    acc = torch.nn.functional.conv_transpose2d(gz.double(), wt.double(), padding=1)
    ref_dx = acc * s.double()[:, :, None, None] + add.double()
    ref_ds = (acc * xin.double()).sum((2, 3))
    gc, aimg, wimg = _v2_operands(gz, wt, None, adjoint=True)
    dx, ds = H.empty_cl(n, ci, h, w, DEV), torch.zeros(n, ci, device=DEV)
    H.conv_v3(aimg, wimg, dx, H.classes_corr_adjoint(h, w, 3, 3, 1), plan=plan, epi=L.EPI_BWD, out_scale=s.to(DEV),
              xin=xin.to(DEV).contiguous(memory_format=torch.channels_last), ds=ds, addend=add.to(DEV).contiguous(memory_format=torch.channels_last), products=products)
    tol = 2e-5 if products == 3 else 2e-3
    close(dx * 1e4, ref_dx.float() * 1e4, tol, 'conv_v3 dgrad dx')
    close(ds * 1e4, ref_ds.float() * 1e4, tol * 2.5, 'conv_v3 dgrad ds')


def test_conv_v3_full_size_layers_deterministic_and_equal_to_conv_v2():
    """The backbone layers the kernel exists for (64^2 x 512 -> 512 and 128^2 x 256 -> 256 at one image) with the fused forward epilogue:
    against fp64, against the pre-split kernel (same products, different summation order: fp32 rounding apart) and bit-identical from
    launch to launch (the K slices of the waves are summed in wave order)."""
    from inv3d_amd import hipops as H, _lib as L
    for (n, ci, h, co) in ((1, 512, 64, 512), (1, 256, 128, 256)):
        w = h
        g = torch.Generator().manual_seed(43)
        x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
        s = (1 + 0.5 * torch.randn(n, ci, generator=g)).to(DEV)
        d = (0.5 + torch.rand(n, co, generator=g)).to(DEV)
        noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
        bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
        aimg = H.split_activation(x, H.absmax(x), in_scale=s)
        wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
        cls = H.classes_corr(h, w, 3, 3, 1)
        z = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1) * d.double()[:, :, None, None]
        ref = (torch.nn.functional.leaky_relu(z + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 1.4).float()
        kw = dict(epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength, act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0)
        o2 = H.empty_cl(n, co, h, w, DEV)
        H.conv_v2(aimg, wimg, o2, cls, patch_rows=4, **kw)
        outs = []
        for plan in ((4, 4), (4, 4), (4, 4), (2, 8), (2, 8)):
            out = H.empty_cl(n, co, h, w, DEV)
            H.conv_v3(aimg, wimg, out, cls, plan=plan, out_amax=torch.zeros(1, device=DEV), **kw)
            torch.cuda.synchronize()
            close(out, ref, 5e-5, f'conv_v3 full size {h} {plan}')
            close(out, o2, 2e-6, f'conv_v3 vs conv_v2 {h} {plan}')
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[3], outs[4])


@pytest.mark.parametrize('rows', [8, 4, 2])
def test_conv_v2_data_gradient_epilogue_vs_torch(rows):
    """Data gradient of a 3x3 layer: adjoint taps on the adjoint weight image, dx = acc * styles + addend, ds = sum_px acc * x."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = 2, 128, 24, 40, 64
    g = torch.Generator().manual_seed(22)
    gz = torch.randn(n, co, h, w, generator=g) * 1e-4                 # gradient-sized operand: the split is range-normalised
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s, xin, add = 1 + 0.5 * torch.randn(n, ci, generator=g), torch.randn(n, ci, h, w, generator=g), torch.randn(n, ci, h, w, generator=g) * 1e-4
    acc = torch.nn.functional.conv_transpose2d(gz.double(), wt.double(), padding=1)
    ref_dx = acc * s.double()[:, :, None, None] + add.double()
    ref_ds = (acc * xin.double()).sum((2, 3))
    gc, aimg, wimg = _v2_operands(gz, wt, None, adjoint=True)
    dx, ds = H.empty_cl(n, ci, h, w, DEV), torch.zeros(n, ci, device=DEV)
    H.conv_v2(aimg, wimg, dx, H.classes_corr_adjoint(h, w, 3, 3, 1), epi=L.EPI_BWD, out_scale=s.to(DEV), xin=xin.to(DEV).contiguous(memory_format=torch.channels_last),
              ds=ds, addend=add.to(DEV).contiguous(memory_format=torch.channels_last), patch_rows=rows)
    close(dx, ref_dx.float(), 2e-5, 'conv_v2 dgrad dx')
    close(ds, ref_ds.float(), 5e-5, 'conv_v2 dgrad ds')


@pytest.mark.parametrize('shape,ks', [((1, 64, 16, 32, 128), 2), ((1, 256, 24, 40, 256), 4), ((2, 128, 9, 33, 128), 8), ((1, 48, 8, 32, 128), 3)])
@pytest.mark.parametrize('products', [3, 1])
def test_conv_v2_split_k_vs_torch(shape, ks, products):
    """Split-K launch of the pre-split kernel (EG3D_EPI_ATOMIC: the 16-channel chunks of the contraction divided over `ks` workgroups per
    tile, partial tiles added with fp32 atomics into a zeroed buffer) -- the 128^2 x 256 / 64^2 x 512 layers, whose grids cannot fill
    the chip -- vs torch fp64, incl. a chunk count that does not divide evenly and the single-product arithmetic."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    ref = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1)
    xc, aimg, wimg = _v2_operands(x, wt, s)
    z = torch.zeros((n, co, h, w), device=DEV).contiguous(memory_format=torch.channels_last)
    H.conv_v2(aimg, wimg, z, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_ATOMIC, ksplit=ks, products=products)
    close(z, ref.float(), 2e-5 if products == 3 else 3e-3, f'conv_v2 split-K {shape} x{ks}')
    one = H.empty_cl(n, co, h, w, DEV)
    H.conv_v2(aimg, wimg, one, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_STORE, products=products)
    close(z, one, 2e-6, 'split-K vs one workgroup per tile (same products, different summation order)')


@pytest.mark.parametrize('shape', [(1, 32, 16, 32, 128), (2, 64, 20, 33, 128)])
def test_conv_v2_transposed_classes_vs_torch(shape):
    """Stride-2 transposed conv as four parity classes (4 / 2 / 2 / 1 taps, three launches) with the plain-store epilogue."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(23)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    # conv2d_resample's up path: transposed conv of the flipped weight = scatter with w[ky,kx]; classes_convT follows the unflipped pack
    ref = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], wt.double().transpose(0, 1), stride=2)
    xc, aimg, wimg = _v2_operands(x, wt, s)
    cls, hz, wz = H.classes_convT(h, w, 3, 3, 2)
    z = H.empty_cl(n, co, hz, wz, DEV)
    H.conv_v2(aimg, wimg, z, cls, out_stride=2, epi=L.EPI_STORE)
    close(z, ref.float(), 2e-5, f'conv_v2 convT {shape}')


@pytest.mark.parametrize('shape,ks', [((1, 64, 16, 32, 64), 1), ((2, 32, 20, 33, 128), 1), ((1, 128, 8, 64, 64), 4), ((1, 48, 19, 40, 192), 3)])
@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize('rows', [8, 4])          # 4: conv_v2_up2r_kernel (four waves, tap-row weight ring, two workgroups per CU: round 6)
def test_conv_up2_fused_parity_vs_torch(shape, ks, products, rows):
    """The fused-parity transposed-conv kernel (csrc/conv_v2_up.hip): all four output parities of the stride-2 3x3 transposed conv from
    one workgroup per 8 x 32 input patch, (i) on the full ragged (Hi + 1) x (Wi + 1) cell grid and (ii) as the model runs it: main grid
    Hi x Wi + the last output row / column as four tap classes of the loader-split kernel; with and without split-K, both arithmetics."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g = torch.Generator().manual_seed(41)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    ref = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], wt.double().transpose(0, 1), stride=2)
    xc, aimg, wimg = _v2_operands(x, wt, s)
    hz, wz = 2 * h + 1, 2 * w + 1
    tol = 2e-5 if products == 3 else 3e-3
    epi = L.EPI_ATOMIC if ks > 1 else L.EPI_STORE
    mk = (lambda: torch.zeros((n, co, hz, wz), device=DEV).contiguous(memory_format=torch.channels_last)) if ks > 1 else (lambda: H.empty_cl(n, co, hz, wz, DEV))
    z = mk()
    H.conv_up2(aimg, wimg, z, epi=epi, ksplit=ks, products=products, patch_rows=rows)
    close(z, ref.float(), tol, f'conv_up2 full grid {shape} x{ks}')
    z2 = mk()
    if ks == 1:
        z2.fill_(float('nan'))             # every output pixel must be written by exactly one of the two launches
    H.conv_up2(aimg, wimg, z2, Hc=h, Wc=w, epi=epi, ksplit=ks, products=products, patch_rows=rows)
    H.conv_igemm(xc, H.pack_weight_fwd(wt.to(DEV)), ci, co, z2, H.up2_border_classes(h, w), out_stride=2, in_scale=s.to(DEV), epi=L.EPI_STORE, precision='f16x3')
    close(z2, ref.float(), tol, f'conv_up2 main grid + border classes {shape} x{ks}')


def test_conv_up2_patch_heights_are_bit_identical():
    """conv_v2_up2_kernel (8 x 32 cells, eight waves) and conv_v2_up2r_kernel (4 x 32 cells, four waves, tap-row weight ring) accumulate every output in the same
    order -- chunk-major, taps in (ky, kx) order -- so their results must be equal bit for bit, on a ragged grid, at the backbone's b256 conv0 shape and with the
    single-product arithmetic."""
    from inv3d_amd import hipops as H, _lib as L
    for (n, ci, h, w, co, products) in ((1, 256, 128, 128, 128, 3), (2, 48, 19, 40, 192, 3), (1, 64, 16, 32, 64, 1)):
        g = torch.Generator().manual_seed(77)
        x = torch.randn(n, ci, h, w, generator=g)
        wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
        s = 1 + 0.5 * torch.randn(n, ci, generator=g)
        xc, aimg, wimg = _v2_operands(x, wt, s)
        outs = []
        for rows in (8, 4):
            z = H.empty_cl(n, co, 2 * h + 1, 2 * w + 1, DEV)
            z.fill_(float('nan'))
            H.conv_up2(aimg, wimg, z, epi=L.EPI_STORE, products=products, patch_rows=rows)
            outs.append(z)
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]), (n, ci, h, w, co, products)


@pytest.mark.parametrize('shape', [(1, 128, 16, 32, 64), (2, 256, 9, 33, 128), (1, 128, 24, 40, 192)])
@pytest.mark.parametrize('products', [3, 1])
def test_conv_v2_stride2_adjoint_vs_torch(shape, products):
    """Data gradient of an up-sampling layer on the parity-split kernel (csrc/conv_v2_s2adj.hip): FIR adjoint + operand split in one pass
    (eg3d_fir44_adjoint_split), then dx = (stride-2 correlation of G with the layer's weights) * styles, ds = sum_px acc * x -- vs torch
    fp64, on ragged grids (Hi, Wi not multiples of the 8 x 32 patch), batch 2, two 128-channel tiles."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape                       # layer: ci -> co, input h x w, output 2h x 2w
    g_ = torch.Generator().manual_seed(51)
    dz = torch.randn(n, co, 2 * h, 2 * w, generator=g_) * 1e-3
    wt = torch.randn(co, ci, 3, 3, generator=g_) / math.sqrt(ci * 9)
    s, xin = 1 + 0.5 * torch.randn(n, ci, generator=g_), torch.randn(n, ci, h, w, generator=g_)
    f1 = torch.tensor([1., 3., 3., 1.], dtype=torch.float64) / 8
    f2 = torch.outer(f1, f1)[None, None].repeat(co, 1, 1, 1)
    G = torch.nn.functional.conv2d(torch.nn.functional.pad(dz.double(), (2, 2, 2, 2)), f2, groups=co) * 4.0              # (2h + 1) x (2w + 1)
    acc = torch.nn.functional.conv2d(G, wt.double().transpose(0, 1), stride=2)
    assert acc.shape == (n, ci, h, w)
    ref_dx, ref_ds = acc * s.double()[:, :, None, None], (acc * xin.double()).sum((2, 3))
    dzc = dz.to(DEV).contiguous(memory_format=torch.channels_last)
    gimg = H.fir44_adjoint_split(dzc, H.absmax(dzc), gain=4.0)
    wimg = H.split_weight(H.pack_weight_adj(wt.to(DEV)), ci, co, 9)
    dx, ds = H.empty_cl(n, ci, h, w, DEV), torch.zeros(n, ci, device=DEV)
    H.conv_v2_s2adj(gimg, wimg, dx, H.classes_convT_adjoint(h, w, 3, 3, 2), epi=L.EPI_BWD, out_scale=s.to(DEV),
                    xin=xin.to(DEV).contiguous(memory_format=torch.channels_last), ds=ds, products=products)
    tol = 2e-5 if products == 3 else 3e-3
    scale = float(ref_dx.abs().max())
    assert float((dx.double().cpu() - ref_dx).abs().max()) <= tol * scale
    assert float((ds.double().cpu() - ref_ds).abs().max()) <= 3 * tol * float(ref_ds.abs().max())
    # the same operand through the loader-split kernel (fp32 G from the stand-alone FIR adjoint): the path this one replaces
    Gf = H.upfirdn2d_nhwc(dzc, torch.outer(f1, f1).float().to(DEV).contiguous(), pad=(2, 2, 2, 2), flip=True, gain=4.0)
    dx2 = H.empty_cl(n, ci, h, w, DEV)
    H.conv_igemm(Gf, H.pack_weight_adj(wt.to(DEV)), co, ci, dx2, H.classes_convT_adjoint(h, w, 3, 3, 2), in_stride=2, epi=L.EPI_BWD, out_scale=s.to(DEV),
                 precision='f16x3', a_amax=H.absmax(dzc), a_amax_mul=4.0)
    if products == 3:
        assert float((dx - dx2).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize('shape', [(1, 128, 16, 32, 64), (2, 256, 9, 33, 128), (1, 64, 24, 40, 192), (1, 512, 32, 32, 64), (1, 64, 5, 7, 64)])
@pytest.mark.parametrize('products', [3, 1])
def test_conv_v3_stride2_adjoint_vs_torch(shape, products):
    """The same data gradient on the wave-split form (conv_v3_s2adj_kernel: the (parity, chunk) items of the contraction dealt to the four waves
    in runs of equal cost): vs torch fp64 on ragged grids, batch 2, 1 .. 8 channel tiles of 64, chunk counts 4, 8, 12."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g_ = torch.Generator().manual_seed(52)
    dz = torch.randn(n, co, 2 * h, 2 * w, generator=g_) * 1e-3
    wt = torch.randn(co, ci, 3, 3, generator=g_) / math.sqrt(ci * 9)
    s, xin = 1 + 0.5 * torch.randn(n, ci, generator=g_), torch.randn(n, ci, h, w, generator=g_)
    f1 = torch.tensor([1., 3., 3., 1.], dtype=torch.float64) / 8
    f2 = torch.outer(f1, f1)[None, None].repeat(co, 1, 1, 1)
    G = torch.nn.functional.conv2d(torch.nn.functional.pad(dz.double(), (2, 2, 2, 2)), f2, groups=co) * 4.0
    acc = torch.nn.functional.conv2d(G, wt.double().transpose(0, 1), stride=2)
    ref_dx, ref_ds = acc * s.double()[:, :, None, None], (acc * xin.double()).sum((2, 3))
    dzc = dz.to(DEV).contiguous(memory_format=torch.channels_last)
    gimg = H.fir44_adjoint_split(dzc, H.absmax(dzc), gain=4.0)
    wimg = H.split_weight(H.pack_weight_adj(wt.to(DEV)), ci, co, 9)
    outs = []
    for rep in range(2):
        dx, ds = H.empty_cl(n, ci, h, w, DEV), torch.zeros(n, ci, device=DEV)
        H.conv_v2_s2adj(gimg, wimg, dx, H.classes_convT_adjoint(h, w, 3, 3, 2), epi=L.EPI_BWD, out_scale=s.to(DEV),
                        xin=xin.to(DEV).contiguous(memory_format=torch.channels_last), ds=ds, products=products, v3=True)
        outs.append(dx)
    tol = 2e-5 if products == 3 else 3e-3
    assert torch.isfinite(dx).all()
    assert float((dx.double().cpu() - ref_dx).abs().max()) <= tol * float(ref_dx.abs().max())
    assert float((ds.double().cpu() - ref_ds).abs().max()) <= 3 * tol * float(ref_ds.abs().max())
    assert torch.equal(outs[0], outs[1])            # the K slices are summed in wave order: run-to-run identical


@pytest.mark.parametrize('shape', [(1, 64, 16, 32, 64), (2, 128, 9, 33, 64), (1, 64, 40, 70, 128), (1, 256, 32, 32, 128)])
@pytest.mark.parametrize('products', [3, 1])
def test_conv_wgrad_v2_up_vs_torch(shape, products):
    """Weight gradient of an up-sampling layer (stride-2 3x3 transposed conv) from the parity-split image of its gradient operand
    (eg3d_fir44_adjoint_split) and the split image of its modulated input (conv_wgrad_v2_up_kernel), vs torch fp64 autograd of F.conv_transpose2d
    on the same FIR-adjoint gradient: ragged strips and row groups, batch 2, several channel tiles; and vs the loader-split kernel it replaces."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = shape
    g_ = torch.Generator().manual_seed(71)
    dz = torch.randn(n, co, 2 * h, 2 * w, generator=g_) * 1e-3
    x = torch.randn(n, ci, h, w, generator=g_)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g_)
    f1 = torch.tensor([1., 3., 3., 1.], dtype=torch.float64) / 8
    f2 = torch.outer(f1, f1)[None, None].repeat(co, 1, 1, 1)
    G = torch.nn.functional.conv2d(torch.nn.functional.pad(dz.double(), (2, 2, 2, 2)), f2, groups=co) * 4.0              # (2h + 1) x (2w + 1)
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    z = torch.nn.functional.conv_transpose2d(x.double() * s.double()[:, :, None, None], wt.transpose(0, 1), stride=2)
    ref, = torch.autograd.grad(z, wt, G)                                   # [co, ci, 3, 3]
    dzc = dz.to(DEV).contiguous(memory_format=torch.channels_last)
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    gimg = H.fir44_adjoint_split(dzc, H.absmax(dzc), gain=4.0)
    ximg = H.split_activation(xc, H.absmax(xc), in_scale=s.to(DEV))
    cls_w = H.classes_convT(h, w, 3, 3, 2)[0]
    wtaps = [0] * 9
    for c in cls_w:
        for t in range(c.ntaps):
            ky, kx = c.out_py - 2 * c.dy[t], c.out_px - 2 * c.dx[t]
            wtaps[3 * ky + kx] = c.wtap[t]
    dwp = torch.zeros(co, 9 * ci, device=DEV)
    H.conv_wgrad_v2_up(gimg, ximg, dwp, wtaps, products=products)
    got = dwp.view(co, 9, ci).permute(0, 2, 1).reshape(co, ci, 3, 3)
    tol = 2e-5 if products == 3 else 3e-3
    scale = float(ref.abs().max())
    assert float((got.double().cpu() - ref).abs().max()) <= tol * scale, float((got.double().cpu() - ref).abs().max()) / scale
    if products == 3:                   # the launch it replaces: fp32 G from the stand-alone FIR adjoint through the loader-split kernel
        Gf = H.upfirdn2d_nhwc(dzc, torch.outer(f1, f1).float().to(DEV).contiguous(), pad=(2, 2, 2, 2), flip=True, gain=4.0)
        dw2 = torch.zeros(co, 9 * ci, device=DEV)
        H.conv_wgrad(xc, Gf, ci, co, dw2, cls_w, in_stride=1, out_stride=2, in_scale=s.to(DEV), precision='f16x3', g_amax=H.absmax(dzc), g_amax_mul=4.0)
        assert float((dwp - dw2).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize('shape', [(1, 64, 16, 32), (2, 128, 9, 20), (1, 192, 33, 17)])
def test_upconv_epilogue_lds_and_fused_split(shape):
    """The LDS-staged separable FIR epilogue of the up layers (eg3d_upconv_epilogue_fwd) vs the 25-load kernel it replaces, on ragged tiles;
    and its fused operand split: the image it writes for the consumer (range bound = conv_clamp, styles folded in) drives the pre-split
    conv to the same result as the stand-alone split pass."""
    from inv3d_amd import hipops as H, _lib as L
    from inv3d_amd.fused import fir44
    n, c, h, w = shape                                  # output h x w, z (h + 1) x (w + 1)
    g = torch.Generator().manual_seed(61)
    z = H.to_cl((torch.randn(n, c, h + 1, w + 1, generator=g) * 3).to(DEV))
    d = (0.5 + torch.rand(n, c, generator=g)).to(DEV)
    noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(c, generator=g)).to(DEV)
    kw = dict(d=d, noise=noise, noise_nstride=0, noise_strength=strength, bias=bias, act='lrelu', alpha=0.2, gain=math.sqrt(2.0))
    for clamp in (-1.0, 4.0):
        ref, out = H.empty_cl(n, c, h, w, DEV), H.empty_cl(n, c, h, w, DEV)
        a0, a1 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
        H.epilogue_fwd(z, ref, fir=fir44(DEV), pad0=1, fir_gain=4.0, clamp=clamp, out_amax=a0, **kw)
        assert H.upconv_epilogue_fwd(z, out, pad0=1, fir_gain=4.0, clamp=clamp, out_amax=a1, **kw) is None
        close(out, ref, 2e-6, f'upconv epilogue {shape} clamp {clamp}')
        assert abs(float(a0) - float(a1)) <= 2e-6 * float(a0)
    if c % 128 == 0:
        s = (1 + 0.5 * torch.randn(n, c, generator=g)).to(DEV)
        out = H.empty_cl(n, c, h, w, DEV)
        simg = H.upconv_epilogue_fwd(z, out, pad0=1, fir_gain=4.0, clamp=4.0, split_in_scale=s, **kw)
        wt = (torch.randn(128, c, 3, 3, generator=g) / math.sqrt(c * 9)).to(DEV)
        wimg = H.split_weight(H.pack_weight_fwd(wt), 128, c, 9)
        y1, y2 = H.empty_cl(n, 128, h, w, DEV), H.empty_cl(n, 128, h, w, DEV)
        H.conv_v2(simg, wimg, y1, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_STORE)
        H.conv_v2(H.split_activation(out, H.absmax(out), in_scale=s), wimg, y2, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_STORE)
        refy = torch.nn.functional.conv2d(out.double().cpu() * s.double().cpu()[:, :, None, None], wt.double().cpu(), padding=1)
        close(y1, refy.float(), 2e-5, 'conv on the image written by the epilogue')
        close(y1, y2, 1e-5, 'fused split vs stand-alone split pass')


def test_conv_v2_heavy_tailed_operands():
    """Operand ranges of a trained network nobody has loaded here: per-channel weight scales e^{N(0,3)}, activations with
    outliers up to 1e6 (beyond the fp16 range: 65 504), styles up to 50.  The split image is range-normalised by max|x| * max|s|, so
    nothing saturates, and the error stays at the level of the exact-fp32 MFMA path, relative to the largest output."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = 1, 128, 32, 32, 128
    g = torch.Generator().manual_seed(24)
    x = torch.randn(n, ci, h, w, generator=g) * torch.exp(torch.randn(1, ci, 1, 1, generator=g) * 2)
    x.view(-1)[torch.randint(0, x.numel(), (64,), generator=g)] = 1e6 * torch.randn(64, generator=g).sign()
    wt = torch.randn(co, ci, 3, 3, generator=g) * torch.exp(torch.randn(1, ci, 1, 1, generator=g) * 3) / 30
    s = torch.randn(n, ci, generator=g) * 10
    s[0, :4] = torch.tensor([50., -50., 45., 30.])
    ref = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1)
    xc, aimg, wimg = _v2_operands(x, wt, s)
    out = H.empty_cl(n, co, h, w, DEV)
    H.conv_v2(aimg, wimg, out, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_STORE)
    out32 = H.empty_cl(n, co, h, w, DEV)
    H.conv_igemm(xc, H.pack_weight_fwd(wt.to(DEV)), ci, co, out32, H.classes_corr(h, w, 3, 3, 1), in_scale=s.to(DEV), precision='f32')
    scale = float(ref.abs().max())
    e_v2, e_32 = float((out.double().cpu() - ref).abs().max()) / scale, float((out32.double().cpu() - ref).abs().max()) / scale
    assert torch.isfinite(out).all() and e_v2 <= 2.0 * e_32 + 1e-7, (e_v2, e_32)
    # tiny operands (gradient-like, 1e-9): same statement
    xs = torch.randn(n, ci, h, w, generator=g) * 1e-9
    ref = torch.nn.functional.conv2d(xs.double(), wt.double(), padding=1)
    xc, aimg, wimg = _v2_operands(xs, wt, None)
    H.conv_v2(aimg, wimg, out, H.classes_corr(h, w, 3, 3, 1), epi=L.EPI_STORE)
    H.conv_igemm(xc, H.pack_weight_fwd(wt.to(DEV)), ci, co, out32, H.classes_corr(h, w, 3, 3, 1), precision='f32')
    scale = float(ref.abs().max())
    e_v2, e_32 = float((out.double().cpu() - ref).abs().max()) / scale, float((out32.double().cpu() - ref).abs().max()) / scale
    assert e_v2 <= 2.0 * e_32 + 1e-7, (e_v2, e_32)


@pytest.mark.parametrize('n', [1, 3, 4, 1023, 147456, 1 << 20, (1 << 20) + 3])
def test_absmax_any_length(n):
    """eg3d_absmax: max|x| of a dense fp32 array (operand range of the split images), any length, non-finite entries ignored."""
    from inv3d_amd import hipops as H
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    x[n // 2] = -7.5
    got = H.absmax(x.to(DEV))
    assert float(got) == 7.5
    if n > 8:
        x[1] = float('inf')
        x[n - 1] = 9.25
        assert float(H.absmax(x.to(DEV))) == 9.25


@pytest.mark.parametrize('kind', ['v2_3x3', 'v3_3x3', 'v3_3x3_rows2_waves8', 'igemm_3x3', 'igemm_1x1', 'igemm_clamp_shared_noise', 'torgb4_elementwise', 'torgb_small', 'torgb_small_clamp_shared_noise',
                                  'torgb_mid', 'torgb_mid_clamp_shared_noise'])
def test_fused_activation_backward_equals_separate_pass(kind):
    """EG3D_EPI_BWD_ACT: a data-gradient launch that also runs the activation backward of the layer that produced its `xin`
    (dz, dbias, dd, dnoise, dstrength, max|dz|) against the two-pass form it replaces (EPI_BWD, then eg3d_modconv_epilogue_bwd on
    the stored dout) -- and the two-pass form against a float64 evaluation of the same expressions."""
    from inv3d_amd import hipops as H, _lib as L
    CL = torch.channels_last

    def close(a, b, tol, what):            # relative to the largest reference value (the operands here are gradient-sized)
        a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
        assert a.shape == b.shape and torch.isfinite(a).all(), what
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert err <= tol * scale, f'{what}: err {err:.3e} > {tol} * {scale:.3e}'

    mid = kind.startswith('torgb_mid')                   # ... its streaming form: 2 x 4608 pixels, 144 tiles per image
    small = kind.startswith('torgb_small') or mid        # csrc/torgb_small.hip: 96 outputs, 63 pixels per image (a ragged last tile)
    one = kind in ('igemm_1x1', 'torgb4_elementwise') or small
    n, ci, h, w, co = (2, 128, 24, 64, 128) if not one else ((2, 128, 32, 32, 4) if not small else ((2, 160, 72, 64, 96) if mid else (2, 160, 9, 7, 96)))
    k = 1 if one else 3
    g = torch.Generator().manual_seed(31)
    gz = torch.randn(n, co, h, w, generator=g) * 1e-3
    wt = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
    s = 1 + 0.5 * torch.randn(n, ci, generator=g)
    xin = torch.randn(n, ci, h, w, generator=g) * 1.5                   # the producing layer's saved output (lrelu'd, gained)
    add = torch.randn(n, ci, h, w, generator=g) * 1e-3
    d = 0.5 + torch.rand(n, ci, generator=g)
    bias = torch.randn(ci, generator=g) * 0.1
    shared = kind.endswith('clamp_shared_noise')
    noise = torch.randn(h, w, generator=g) if shared else torch.randn(n, 1, h, w, generator=g)
    strength = torch.tensor(0.37)
    gain, clamp, alpha = math.sqrt(2), (1.2 if shared else -1.0), 0.2
    dev = lambda t: t.to(DEV).contiguous(memory_format=CL) if t.dim() == 4 and t.shape[1] > 1 else t.to(DEV).contiguous()
    xin_d, add_d, s_d, d_d, b_d, nz_d, st_d = dev(xin), dev(add), dev(s), dev(d), dev(bias), dev(noise), strength.to(DEV)
    nstride = 0 if shared else h * w
    cls = H.classes_corr_adjoint(h, w, k, k, k // 2)

    def launch(spec, out_amax):
        dx, ds = H.empty_cl(n, ci, h, w, DEV), torch.zeros(n, ci, device=DEV)
        kw = dict(epi=L.EPI_BWD, out_scale=s_d, xin=xin_d, ds=ds, addend=add_d, out_amax=out_amax)
        if spec is not None:
            kw['act_bwd'] = spec
        if kind == 'v2_3x3':
            _, aimg, wimg = _v2_operands(gz, wt, None, adjoint=True)
            r = H.conv_v2(aimg, wimg, dx, cls, **kw)
        elif kind.startswith('v3_3x3'):
            _, aimg, wimg = _v2_operands(gz, wt, None, adjoint=True)
            r = H.conv_v3(aimg, wimg, dx, cls, plan=(2, 8) if kind.endswith('waves8') else (4, 4), **kw)
        else:
            wa = wt.permute(1, 2, 3, 0).reshape(ci, k * k * co).to(DEV).contiguous()          # adjoint pack [I][tap][O]
            if kind == 'torgb4_elementwise' and spec is not None:          # the element-wise form of the same launch (eg3d_torgb_dgrad_act)
                H.torgb_dgrad_act(dev(gz), wa, xin_d, s_d, dx, spec, ds=ds, addend=add_d, dz_amax=out_amax)
                return dx, ds, True
            if small:
                r = H.torgb_small_bwd(dev(gz), wa, s_d, xin_d, dx, ds=ds, addend=add_d, act_bwd=spec, out_amax=out_amax)
                assert r is not None, 'the small launch refused its own geometry'
                return dx, ds, r
            r = H.conv_igemm(dev(gz), wa, co, ci, dx, cls, precision='bf16x6', **kw)
        return dx, ds, r

    def accs():
        return dict(dbias=torch.zeros(ci, device=DEV), dd=torch.zeros(n, ci, device=DEV), dnoise=torch.zeros_like(nz_d), dstrength=torch.zeros((), device=DEV))

    # two passes
    dout, ds_a, _ = launch(None, None)
    A = accs()
    amax_a = torch.zeros(1, device=DEV)
    dz_a = H.epilogue_bwd(dout, xin_d, H.empty_cl(n, ci, h, w, DEV), d=d_d, noise=nz_d, noise_nstride=nstride, noise_strength=st_d, bias=b_d, act='lrelu',
                          alpha=alpha, gain=gain, clamp=clamp, dnoise_nstride=nstride, dz_amax=amax_a, **A)
    # one pass
    B = accs()
    amax_b = torch.zeros(1, device=DEV)
    spec = H.ActBwdSpec(d=d_d, bias=b_d, noise=nz_d, noise_nstride=nstride, noise_strength=st_d, act='lrelu', alpha=alpha, gain=gain, clamp=clamp,
                        dnoise_nstride=nstride, **B)
    dz_b, ds_b, fused = launch(spec, amax_b)
    assert fused is True, 'the fused epilogue was not taken'
    close(dz_b, dz_a, 1e-6, f'{kind} dz')
    close(ds_b, ds_a, 1e-5, f'{kind} ds')
    for key in A:
        close(B[key], A[key], 2e-5, f'{kind} {key}')
    assert abs(float(amax_b) - float(dz_a.abs().max())) <= 1e-6 * float(amax_b)
    # float64 statement of the activation backward on the stored dout
    o, do = xin.double(), dout.cpu().double()
    yy = o / gain
    dy = do * gain * torch.where(yy > 0, 1.0, alpha)
    if clamp >= 0:
        dy = torch.where(o.abs() >= clamp, 0.0, dy)
    pre = torch.where(yy > 0, yy, yy / alpha)
    nz64 = (noise.double() * float(strength)).reshape((1, 1, h, w) if shared else (n, 1, h, w))
    close(dz_b, (dy * d.double()[:, :, None, None]).float(), 1e-5, f'{kind} dz vs f64')
    close(B['dbias'], dy.sum((0, 2, 3)).float(), 5e-5, f'{kind} dbias vs f64')
    close(B['dd'], ((dy * (pre - bias.double()[None, :, None, None] - nz64)).sum((2, 3)) / d.double()).float(), 5e-5, f'{kind} dd vs f64')
    dn = dy.sum(1, keepdim=True) * float(strength)
    close(B['dnoise'], (dn.sum(0)[0] if shared else dn).float(), 5e-5, f'{kind} dnoise vs f64')
    close(B['dstrength'], (dy.sum(1, keepdim=True) * noise.double().reshape(nz64.shape)).sum().float(), 1e-4, f'{kind} dstrength vs f64')


@pytest.mark.parametrize('with_addend', [False, True])
def test_torgb_dgrad_act_split_image_decodes_to_dz(with_addend):
    """eg3d_torgb_dgrad_act_split: the same pass as eg3d_torgb_dgrad_act, dz written as the two-piece fp16 operand image (range from a bound
    on max|dz|, no split pass).  The image decodes to the fp32 dz within the two-piece resolution, the scale covers max|dz| with a bound at
    most 2^8 loose, the reductions are those of the fp32 form, and the data gradient run on the image equals the one run on the split of dz."""
    from inv3d_amd import hipops as H, _lib as L
    CL = torch.channels_last
    n, c, h, w = 2, 128, 40, 48
    g = torch.Generator().manual_seed(77)
    dy4 = (torch.randn(n, 4, h, w, generator=g) * 1e-3).to(DEV).contiguous(memory_format=CL)
    wa = (torch.randn(c, 4, generator=g) / 2).to(DEV).contiguous()
    s = (1 + 0.5 * torch.randn(n, c, generator=g)).to(DEV)
    x = (torch.randn(n, c, h, w, generator=g) * 1.5).to(DEV).contiguous(memory_format=CL)
    add = (torch.randn(n, c, h, w, generator=g) * 2e-3).to(DEV).contiguous(memory_format=CL) if with_addend else None
    d = (0.5 + torch.rand(n, c, generator=g)).to(DEV)
    bias = (torch.randn(c, generator=g) * 0.1).to(DEV)
    noise = torch.randn(n, 1, h, w, generator=g).to(DEV)
    st = torch.tensor(0.37, device=DEV)

    def spec_and_accs():
        A = dict(dbias=torch.zeros(c, device=DEV), dd=torch.zeros(n, c, device=DEV), dnoise=torch.zeros_like(noise), dstrength=torch.zeros((), device=DEV))
        return H.ActBwdSpec(d=d, bias=bias, noise=noise, noise_nstride=h * w, noise_strength=st, act='lrelu', alpha=0.2, gain=math.sqrt(2), clamp=256.0,
                            dnoise_nstride=h * w, **A), A
    spec_a, A = spec_and_accs()
    ds_a, amax_a = torch.zeros(n, c, device=DEV), torch.zeros(1, device=DEV)
    dz = H.torgb_dgrad_act(dy4, wa, x, s, H.empty_cl(n, c, h, w, DEV), spec_a, ds=ds_a, addend=add, dz_amax=amax_a)
    spec_b, B = spec_and_accs()
    ds_b = torch.zeros(n, c, device=DEV)
    simg = H.torgb_dgrad_act_split(dy4, wa, x, s, spec_b, H.absmax(dy4), ds=ds_b, addend=add, addend_amax=H.absmax(add) if add is not None else None)
    torch.cuda.synchronize()
    scale = float(simg.scale)
    assert scale > 0 and math.log2(scale) == int(math.log2(scale))
    top = float(amax_a) * scale
    assert 2.0 ** 5 <= top < 2.0 ** 14, f'max|dz| * scale = {top}: the bound is wrong or useless'
    img = simg.data.view(n, 2, c // 8, h, w, 8).float()
    dec = ((img[:, 0] + img[:, 1] / 2048.0) / scale).permute(0, 1, 4, 2, 3).reshape(n, c, h, w)       # [n][octet][8][h][w] -> channels
    err = float((dec - dz).abs().max())
    assert err <= 2.0 ** -21 * float(amax_a) * 2.0 ** (14 - math.floor(math.log2(top))) + 1e-30, err
    for key in A:
        close(B[key], A[key], 1e-6, key)
    close(ds_b, ds_a, 1e-6, 'ds')
    # the consumer: a 3x3 data gradient on the image vs on the split of the fp32 dz
    wt = torch.randn(c, c, 3, 3, generator=g) / math.sqrt(c * 9)
    wimg = H.split_weight(H.pack_weight_fwd(wt.to(DEV)), c, c, 9)
    o1, o2 = H.empty_cl(n, c, h, w, DEV), H.empty_cl(n, c, h, w, DEV)
    H.conv_v2(simg, wimg, o1, H.classes_corr(h, w, 3, 3, 1))
    H.conv_v2(H.split_activation(dz, amax_a), wimg, o2, H.classes_corr(h, w, 3, 3, 1))
    close(o1, o2, 2e-6, 'conv on the fused image vs on the split pass')


@pytest.mark.parametrize('shape', [(1, 512, 8, 8), (2, 256, 32, 32), (1, 96, 16, 16)])
def test_dgrad_finish_with_activation_backward_equals_two_passes(shape):
    """eg3d_dgrad_finish_act (split-K layers) against eg3d_dgrad_finish followed by eg3d_modconv_epilogue_bwd."""
    from inv3d_amd import hipops as H
    n, c, h, w = shape
    g = torch.Generator().manual_seed(41)
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    z, x, add = cl(torch.randn(n, c, h, w, generator=g) * 1e-3), cl(torch.randn(n, c, h, w, generator=g)), cl(torch.randn(n, c, h, w, generator=g) * 1e-3)
    s, d, bias = (1 + 0.5 * torch.randn(n, c, generator=g)).to(DEV), (0.5 + torch.rand(n, c, generator=g)).to(DEV), (torch.randn(c, generator=g) * 0.1).to(DEV)
    noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.4, device=DEV)
    gain = math.sqrt(2)

    def accs():
        return dict(dbias=torch.zeros(c, device=DEV), dd=torch.zeros(n, c, device=DEV), dnoise=torch.zeros_like(noise), dstrength=torch.zeros((), device=DEV))
    A, B = accs(), accs()
    ds_a, ds_b = torch.zeros(n, c, device=DEV), torch.zeros(n, c, device=DEV)
    am_a, am_b = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    dout = H.dgrad_finish(z, x, s, H.empty_cl(n, c, h, w, DEV), ds=ds_a, addend=add)
    dz_a = H.epilogue_bwd(dout, x, H.empty_cl(n, c, h, w, DEV), d=d, noise=noise, noise_strength=strength, bias=bias, act='lrelu', alpha=0.2, gain=gain,
                          dz_amax=am_a, **A)
    spec = H.ActBwdSpec(d=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength, act='lrelu', alpha=0.2, gain=gain, clamp=-1.0,
                        dnoise_nstride=0, **B)
    dz_b = H.dgrad_finish_act(z, x, s, H.empty_cl(n, c, h, w, DEV), spec, ds=ds_b, addend=add, dz_amax=am_b)

    def rel(a, b, tol, what):
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert err <= tol * scale, f'{what}: {err:.3e} > {tol} * {scale:.3e}'
    rel(dz_b, dz_a, 1e-6, 'dz')
    rel(ds_b, ds_a, 1e-5, 'ds')
    for k in A:
        rel(B[k], A[k], 2e-5, k)
    assert float(am_b) == float(am_a)


def test_conv_single_product_fp16_arithmetic():
    """EG3D_PREC_F16X1 (the reference's fp16 layers): one product of fp16-rounded, range-normalised operands, fp32 accumulation.  Error
    against float64 at the level of fp16 operand rounding (2^-11 per operand), on the pre-split kernel and the weight gradient; against the exact product of the ROUNDED operands it is fp32-accumulation sized."""
    from inv3d_amd import hipops as H, _lib as L
    n, ci, h, w, co = 1, 128, 64, 64, 128
    g = torch.Generator().manual_seed(51)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    s = 1 + 0.3 * torch.randn(n, ci, generator=g)
    ref = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1)
    xc, aimg, wimg = _v2_operands(x, wt, s)
    cls = H.classes_corr(h, w, 3, 3, 1)
    out1, out3 = H.empty_cl(n, co, h, w, DEV), H.empty_cl(n, co, h, w, DEV)
    H.conv_v2(aimg, wimg, out1, cls, epi=L.EPI_STORE, products=1)
    H.conv_v2(aimg, wimg, out3, cls, epi=L.EPI_STORE, products=3)
    scale = float(ref.abs().max())
    e1, e3 = float((out1.cpu().double() - ref).abs().max()) / scale, float((out3.cpu().double() - ref).abs().max()) / scale
    assert e3 < 2e-6 and 2e-5 < e1 < 2e-3, (e1, e3)
    wf = wt.permute(0, 2, 3, 1).reshape(co, 9 * ci).to(DEV).contiguous()
    with pytest.raises(L.Eg3dHipError):          # the loader-split kernel declines the mode (its callers keep three products)
        H.conv_igemm(xc, wf, ci, co, H.empty_cl(n, co, h, w, DEV), cls, in_scale=s.to(DEV), epi=L.EPI_STORE, precision='f16x1')
    gz = (torch.randn(n, co, h, w, generator=g) * 1e-3).to(DEV).contiguous(memory_format=torch.channels_last)
    dref = torch.nn.grad.conv2d_weight((x.double() * s.double()[:, :, None, None]), wt.shape, gz.cpu().double(), padding=1)
    for prec, lo, hi in (('f16x3', 0.0, 5e-6), ('f16x1', 1e-5, 2e-3)):
        dwp = torch.zeros(co, 9 * ci, device=DEV)
        H.conv_wgrad(xc, gz, ci, co, dwp, cls, in_scale=s.to(DEV), precision=prec, g_amax=H.absmax(gz))
        got = dwp.view(co, 3, 3, ci).permute(0, 3, 1, 2).cpu().double()
        e = float((got - dref).abs().max()) / float(dref.abs().max())
        assert lo <= e < hi, (prec, e)


def test_batched_weight_pack_equals_per_layer_pack():
    """eg3d_pack_conv_weights_batched (all layers of a network, one launch) writes the same images as the per-layer packs, including the
    zero-padded 3 -> 4 channel toRGB images; eg3d_weight_grad_finish = packed gradient image -> parameter layout + demodulation term."""
    from inv3d_amd import hipops as H
    g = torch.Generator().manual_seed(3)
    shapes = [(32, 16, 3, 3), (16, 32, 3, 3), (3, 16, 1, 1), (96, 24, 1, 1), (40, 8, 3, 3)]
    ws = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    items, outs = [], []
    for w in ws:
        o, i, kh, kw = w.shape
        pad = (o + 3) // 4 * 4 if o % 4 else 0
        wf = torch.zeros((pad or o, kh * kw * i), device=DEV)
        wa = torch.zeros((i, kh * kw * (pad or o)), device=DEV)
        wsq = torch.zeros((o, i), device=DEV)
        items.append((w, wf, wa, wsq, pad))
        outs.append((wf, wa, wsq))
    H.pack_conv_weights_batched(items)
    for w, (wf, wa, wsq) in zip(ws, outs):
        o, i, kh, kw = w.shape
        rf, ra, rq = H.pack_conv_weight(w)
        assert torch.equal(wf[:o], rf) and float(wf[o:].abs().sum()) == 0.0
        ra_p = wa.view(i, kh * kw, -1)
        assert torch.equal(ra_p[:, :, :o].reshape(i, -1), ra) and float(ra_p[:, :, o:].abs().sum()) == 0.0
        assert torch.equal(wsq, rq)
    # gradient finishing pass
    w = ws[0]
    o, i, kh, kw = w.shape
    n = 2
    dwp = torch.randn(o, kh * kw * i, generator=g).to(DEV)
    s, d, dd = torch.randn(n, i, generator=g).to(DEV), torch.rand(n, o, generator=g).to(DEV) + 0.5, torch.randn(n, o, generator=g).to(DEV)
    got = H.weight_grad_finish(dwp, w, s, d, dd)
    dwsq = torch.einsum('no,ni->oi', dd * (-0.5) * d ** 3, s * s)
    ref = dwp.view(o, kh, kw, i).permute(0, 3, 1, 2) + 2 * w * dwsq[:, :, None, None]
    close(got, ref, 1e-5, 'weight_grad_finish')
    close(H.weight_grad_finish(dwp, w, None, None, None), dwp.view(o, kh, kw, i).permute(0, 3, 1, 2), 1e-7, 'weight_grad_finish (no demodulation)')


def test_conv_igemm_presplit_weights_bit_identical():
    """eg3d_conv_params::w_presplit: the loader copies the weight pieces written by eg3d_split_weight_pieces instead of forming them --
    the same bits, so the result must be IDENTICAL to the in-loader split (forward and a strided data-gradient geometry)."""
    from inv3d_amd import hipops as H, _lib as L
    g = torch.Generator().manual_seed(11)
    for (ci, co, h, up) in ((32, 64, 24, 1), (16, 128, 16, 2)):
        x = torch.randn(2, ci, h, h, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, 3, 3, generator=g).to(DEV)
        wf, wa, _ = H.pack_conv_weight(w)
        cls = H.classes_corr(h, h, 3, 3, 1) if up == 1 else H.classes_convT(h, h, 3, 3, up)[0]
        ho = h if up == 1 else cls_out_size(h, up)
        outs = []
        for pieces in (None, H.split_weight_pieces(wf)):
            out = H.zeros_cl(2, co, ho, ho, DEV)
            H.conv_igemm(x, wf, ci, co, out, cls, out_stride=up, epi=L.EPI_STORE, precision='f16x3', w_pieces=pieces)
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1) if up == 1 else None
        if ref is not None:
            close(outs[1], ref, 2e-6 * math.sqrt(ci * 9), 'conv_igemm with pre-split weights')


def cls_out_size(h, up):
    return h * up + 1
