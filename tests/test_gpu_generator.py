"""GPU parity of the whole hot path: TriPlaneGenerator.synthesis forward + backward through the C-ABI kernels vs
(a) the golden vectors from the reference (small generator, both noise modes; full-size probes) and (b) the oracle.
Bar: image PSNR(build, ref) >= 60 dB in fp32; here the observed error is ~1e-5 absolute."""
import math

import numpy as np
import pytest
import torch

from oracle import eg3d_oracle as O

pytestmark = pytest.mark.gpu
from inv3d_amd import _lib as _L
DET = _L.DETERMINISTIC
DEV = 'cuda'


def t(a, dev=DEV):
    return torch.from_numpy(np.asarray(a)).to(dev)


def close(a, b, tol, what=''):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), f'{what}: non-finite values'
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f'{what}: err {err:.3e} > {tol} * {scale:.3e}'


def close_most(a, b, tol, what='', frac=0.002, loose=0.05):
    """Per-ray coordinate gradients are piecewise constant in the sample position (bilinear texel boundaries): a sample
    whose coordinate differs by one ulp between CPU and GPU can flip a floor() and change that ray's gradient by O(1/samples).
    Require the tight tolerance on all but `frac` of the rows and a loose bound on the rest."""
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape and torch.isfinite(a).all(), what
    scale = max(1.0, float(b.abs().max()))
    err = (a - b).abs().reshape(-1, a.shape[-1]).amax(-1)
    bad = int((err > tol * scale).sum())
    assert bad <= max(1, int(frac * err.numel())), f'{what}: {bad}/{err.numel()} rows above {tol}'
    assert float(err.max()) <= loose * scale, f'{what}: worst row {float(err.max()):.3e}'


def psnr(a, b, peak=2.0):
    mse = float(((a.double().cpu() - b.double().cpu()) ** 2).mean())
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


def small_G():
    from inv3d_amd import synthetic as S
    cfg = O.small_config()
    G = S.make_generator(w_dim=32, z_dim=32, plane_res=32, channel_base=256, channel_max=16, nrr=16, sr_in_res=16, sr_widths=(16, 8),
                         rendering_kwargs=cfg.rendering, device=DEV)
    S.load_synthetic_weights(G, 0)
    return cfg, G


@pytest.mark.parametrize('mode', ['const', 'random'])
def test_graph_small_golden(golden, mode):
    d = golden('graph_small')
    cfg, G = small_G()
    n = 2
    ws = t(d['ws']).requires_grad_(True)
    c = t(d['c']).requires_grad_(True)
    u1, u2 = O.make_uniforms(cfg, n, seed=4)
    noises = None
    if mode == 'random':
        noises = {}
        for r in cfg.block_resolutions:
            for conv in (['conv1'] if r == 4 else ['conv0', 'conv1']):
                nm = f'backbone.synthesis.b{r}.{conv}'
                noises[nm] = O._randn('noise.' + nm, 6, (n, 1, r, r)).to(DEV)
    names = ['backbone.synthesis.b8.conv0.weight', 'backbone.synthesis.b16.torgb.weight', 'superresolution.block1.conv1.weight',
             'decoder.net.0.weight', 'backbone.synthesis.b32.conv1.noise_strength', 'backbone.synthesis.b16.conv0.bias',
             'superresolution.block0.conv0.affine.weight']
    pd = dict(G.named_parameters())
    for p in G.parameters():
        p.requires_grad_(False)
    for nm in names:
        pd[nm].requires_grad_(True)
    o = G.synthesis(ws, c, noise_mode=mode, force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)), noise_inject=noises)
    m = mode[0]
    close(o['image'], d[f'{m}_image'], 5e-5, 'image')
    close(o['image_raw'], d[f'{m}_raw'], 5e-5, 'image_raw')
    close(o['image_depth'], d[f'{m}_depth'], 5e-5, 'image_depth')
    assert psnr(o['image'], t(d[f'{m}_image'])) > 80
    g = torch.autograd.grad([o['image'], o['image_raw'], o['image_depth']], [ws, c] + [pd[nm] for nm in names],
                            [t(d['g_img']), t(d['g_raw']), t(d['g_dep'])])
    close(g[0], d[f'{m}_dws'], 5e-4, 'd ws')
    close(g[1], d[f'{m}_dc'], 1e-2, 'd c')      # sum over rays of piecewise-constant coordinate gradients: see close_most
    for nm, gv in zip(names, g[2:]):
        close(gv, d[f'{m}_d.{nm}'], 5e-4, f'd {nm}')


def test_planes_and_cache(golden):
    d = golden('graph_small')
    cfg, G = small_G()
    ws, c = t(d['ws']), t(d['c'])
    with torch.no_grad():
        planes = G.backbone.synthesis(ws, noise_mode='const')
        close(planes, d['c_planes'], 5e-5, 'planes')
        u1, u2 = O.make_uniforms(cfg, 2, seed=4)
        a = G.synthesis(ws, c, noise_mode='const', cache_backbone=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
        b = G.synthesis(ws * 0, c, noise_mode='const', use_cached_backbone=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
        close(b['image_raw'], a['image_raw'], 0, 'cached backbone')
        wmap = G.mapping(t(d['map_z']), c, truncation_psi=0.7, truncation_cutoff=5)
        close(wmap, d['map_out'], 1e-4, 'mapping')


def test_noise_const_gradient():
    """Phase A optimises the noise_const buffers (w_projector.py:103-104,126-131): their gradient vs the oracle."""
    cfg, G = small_G()
    P = O.synth_params(cfg, 0)
    n = 1
    ws = O.synth_ws(cfg, n, seed=1)
    c = O.synth_cameras(n, seed=2)
    u1, u2 = O.make_uniforms(cfg, n, seed=4)
    nb = [k for k in P if k.endswith('noise_const') and k.startswith('backbone')]
    Pg = {k: (v.clone().requires_grad_(True) if k in nb else v) for k, v in P.items()}
    o = O.synthesis(Pg, cfg, ws, c, u1, u2, noise_mode='const')
    gi = O._randn('gi', 1, o['image'].shape)
    gr = torch.autograd.grad(o['image'], [Pg[k] for k in nb], gi)
    bufs = {k: b for k, b in G.named_buffers() if k in nb}
    for b in bufs.values():
        b.requires_grad_(True)
    og = G.synthesis(ws.to(DEV), c.to(DEV), noise_mode='const', render_uniforms=(u1.to(DEV), u2.to(DEV)))
    gg = torch.autograd.grad(og['image'], [bufs[k] for k in nb], gi.to(DEV))
    for k, a, b in zip(nb, gg, gr):
        close(a, b, 5e-4, f'd {k}')


# Full-size d ws / d c against the reference's own class, RELATIVE to the gradient's largest entry.  Measured in round 5 (normal and
# deterministic build, two runs each): d ws 3.8e-4 .. 4.1e-4, d c 1.14e-2 .. 1.25e-2 (absolute 2.8e-5 on entries of 2.4e-3: the camera
# gradient is a sum over 16 384 rays of per-ray terms that are piecewise constant in the sample position -- a sample one ulp across a texel
# boundary changes its ray's term by O(1 / samples); the same in both builds, i.e. not summation order).  Bounds = worst observed x 3.
EWS_BOUND, EC_BOUND = 1.5e-3, 4e-2


def test_graph_full_golden(golden):
    """Full-size ffhqrebalanced512-128-shaped generator vs the reference's own TriPlaneGenerator (probe samples)."""
    from inv3d_amd import synthetic as S
    d = golden('graph_full')
    cfg = O.full_config()
    G = S.make_generator(device=DEV)
    S.load_synthetic_weights(G, 0)
    for p in G.parameters():
        p.requires_grad_(False)
    ws = t(d['ws']).requires_grad_(True)
    c = t(d['c']).requires_grad_(True)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    o = G.synthesis(ws, c, noise_mode='const', force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
    img, raw, dep = o['image'], o['image_raw'], o['image_depth']
    assert img.shape == (1, 3, 512, 512) and raw.shape == (1, 3, 128, 128) and dep.shape == (1, 1, 128, 128)
    ip, rp, dp = img.flatten()[t(d['idx_img'])], raw.flatten()[t(d['idx_raw'])], dep.flatten()[t(d['idx_dep'])]
    close(ip, d['img_probe'], 2e-4, 'image probes')
    close(rp, d['raw_probe'], 2e-4, 'raw probes')
    close(dp, d['dep_probe'], 2e-4, 'depth probes')
    # PSNR on the probe set (peak-to-peak 2): the 60 dB bar of SURVEY section 8c
    assert psnr(ip, t(d['img_probe'])) > 60, psnr(ip, t(d['img_probe']))
    st = lambda x: np.array([float(x.mean()), float(x.abs().mean()), float(x.min()), float(x.max())])
    np.testing.assert_allclose(st(img), d['img_stats'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(st(dep), d['dep_stats'], rtol=0, atol=2e-4)
    g_img = O._randn('gf_img', 8, img.shape) / (3 * 512 * 512)
    g_dep = O._randn('gf_dep', 8, dep.shape) / (128 * 128)
    dws, dc = torch.autograd.grad([img, dep], [ws, c], [g_img.to(DEV), g_dep.to(DEV)])
    rel = lambda a, b: float((a.detach().cpu().double() - t(b).cpu().double()).abs().max() / float(t(b).abs().max()))      # noqa: E731
    e_ws, e_c = rel(dws, d['dws']), rel(dc, d['dc'])
    print(f'full-size gradient errors relative to max|ref|: d ws {e_ws:.2e}, d c {e_c:.2e}')
    # rounds 1-4 held 2e-3 / 1e-2 of max(1, max|ref|) -- with gradients of size 1e-4 that bound said nothing.  Relative to the gradient's own
    # largest entry (both builds, two runs each in round 5: see DESIGN.md section 4)
    assert e_ws <= EWS_BOUND, e_ws
    assert e_c <= EC_BOUND, e_c


def _cond_loss_cotangents(img, dep):
    """The conditioned objective of tests/golden/make_golden.py::cond_loss_cotangents, restated (the generator does not travel):
    L = mean((avg_pool2(image) - target)^2) + 0.1 mean(depth^2), target = tanh(bilinear(N(0,1) 16^2 -> 256^2))."""
    import torch.nn.functional as F
    low = O._randn('cond_target', 11, (1, 3, 16, 16))
    target = torch.tanh(F.interpolate(low, size=(256, 256), mode='bilinear', align_corners=False)).to(img.device)
    img = img.detach().requires_grad_(True)
    dep = dep.detach().requires_grad_(True)
    L = (F.avg_pool2d(img, 2) - target).square().mean() + 0.1 * dep.square().mean()
    return torch.autograd.grad(L, [img, dep])


# observed (round 6, both builds) x 3 -- DESIGN.md section 4
# observed (round 6; normal | deterministic build): plain probes / range <= 2.7e-5 (depth), PSNR >= 103.5 dB, d ws 3.1e-6 | 3.4e-6, d c 7.0e-5 | 2.6e-5;
#                                                   heavy probes / range <= 9.5e-6, PSNR >= 114.6 dB, d ws 4.2e-5 | 4.3e-5, d c 7.1e-5 | 5.4e-5
COND_BOUNDS = dict(plain=dict(probe=8e-5, psnr=95.0, dws=1e-5, dc=2.1e-4), heavy=dict(probe=3e-5, psnr=105.0, dws=1.3e-4, dc=2.2e-4))


@pytest.mark.parametrize('weights', ['plain', 'heavy'])
def test_graph_full_conditioned_and_heavy_tailed(golden, weights):
    """Full-size generator vs the reference's own TriPlaneGenerator under a CONDITIONED cotangent (a smooth image + depth objective: with the white-noise
    cotangent of test_graph_full_golden a 3 % camera-gradient bug passes), (i) on the N(0,1) synthetic weights and (ii) on heavy-tailed ones -- per-channel
    log-normal weight gains, x100 outlier channels in b4.const, x10 affine-bias entries, noise_strength up to 1: the statistics of a trained checkpoint,
    where the f16x3 operand split (one power-of-two range per tensor from max|x| max|style|) loses precision first.  Probes relative to the reference
    output's own range; PSNR > 60 dB on the probe set; d ws, d c relative to the gradient's largest entry."""
    from inv3d_amd import synthetic as S
    d = golden('graph_full_cond')
    cfg = O.full_config()
    G = S.make_generator(device=DEV)
    S.load_synthetic_weights(G, 0)
    if weights == 'heavy':
        S.apply_heavy_tail(G, 0)
        P = O.heavy_tailed_params(O.synth_params(cfg, seed=0), seed=0)          # the product-side transform is the oracle's, tensor for tensor
        sd = G.state_dict()
        for k in ('backbone.synthesis.b4.const', 'backbone.synthesis.b64.conv1.weight', 'superresolution.block1.conv1.weight', 'backbone.synthesis.b256.conv0.affine.bias',
                  'backbone.synthesis.b128.conv1.noise_strength'):
            assert torch.equal(sd[k].cpu(), P[k]), k
    for p in G.parameters():
        p.requires_grad_(False)
    from inv3d_amd import hipops as H
    H.weights_changed()
    ws = t(d['ws']).requires_grad_(True)
    c = t(d['c']).requires_grad_(True)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    o = G.synthesis(ws, c, noise_mode='const', force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
    img, raw, dep = o['image'], o['image_raw'], o['image_depth']
    B = COND_BOUNDS[weights]
    for name, x, ik, pk, sk in (('image', img, 'idx_img', 'img_probe', 'img_stats'), ('raw', raw, 'idx_raw', 'raw_probe', 'raw_stats'), ('depth', dep, 'idx_dep', 'dep_probe', 'dep_stats')):
        ref = t(d[f'{weights}.{pk}'])
        rng = float(d[f'{weights}.{sk}'][3] - d[f'{weights}.{sk}'][2])            # the reference output's own range (heavy tails: far beyond [-1, 1])
        got = x.flatten()[t(d[f'{weights}.{ik}'])]
        assert torch.isfinite(x).all(), name
        err = float((got - ref).abs().max()) / rng
        ps = psnr(got, ref, peak=rng)
        print(f'{weights} {name}: range {rng:.4g}, probe err / range {err:.2e}, PSNR {ps:.1f} dB')
        assert err <= B['probe'], (name, err)
        assert ps > B['psnr'], (name, ps)
    g_img, g_dep = _cond_loss_cotangents(img, dep)
    dws, dc = torch.autograd.grad([img, dep], [ws, c], [g_img, g_dep])
    rel = lambda a, b: float((a.detach().cpu().double() - t(b).cpu().double()).abs().max() / float(t(b).abs().max()))      # noqa: E731
    e_ws, e_c = rel(dws, d[f'{weights}.dws']), rel(dc, d[f'{weights}.dc'])
    print(f'{weights}: conditioned-cotangent gradient errors relative to max|ref|: d ws {e_ws:.2e}, d c {e_c:.2e}')
    assert e_ws <= B['dws'], e_ws
    assert e_c <= B['dc'], e_c


# median probe error of every tensor with >= 64 probes, relative to max|g|: observed (round 6, both builds) <= 1.7e-4 (b64.torgb.bias), 7e-5 and below elsewhere;
# single-product SR head: <= 6e-3.  Bounds = observed x 3.
BULK_BOUND_F16X3, BULK_BOUND_F16X1 = 5e-4, 2e-2


@pytest.mark.parametrize('arith', ['f16x3', 'sr_f16x1'])
def test_graph_full_weight_grads_golden(golden, arith):
    """Phase B at full size (base_coach.py:96-99: Adam over every weight): weight, bias, affine, noise-strength, decoder and noise_const
    gradients of the ffhqrebalanced512-128-shaped generator against probes recorded from the reference's own TriPlaneGenerator
    (tests/golden/make_golden.py::gen_graph_full) -- the weight-gradient GEMMs, the decoder Gram kernels and the style bank at the
    geometries they were tuned for (512^2 x 128, 256^2 x 256, XCD-remapped cell slices).  'f16x3': the fp32-equivalent default, bound
    3e-3 of each tensor's max|g| (see below); 'sr_f16x1': the SR head in the reference's fp16-operand
    arithmetic (one product, what PivotalTuner runs by default) -- a looser, stated bound."""
    from inv3d_amd import synthetic as S
    d = golden('graph_full')
    cfg = O.full_config()
    G = S.make_generator(device=DEV)
    S.load_synthetic_weights(G, 0)
    G.requires_grad_(True)
    wkeys = [k[len('wg_idx.'):] for k in d.files if k.startswith('wg_idx.')]
    assert len(wkeys) >= 20
    named = dict(G.named_parameters())
    bufs = dict(G.named_buffers())
    leaves = []
    for k in wkeys:
        if k in named:
            leaves.append(named[k])
        else:
            leaves.append(bufs[k].requires_grad_(True))
    ws, c = t(d['ws']).requires_grad_(True), t(d['c']).requires_grad_(True)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    kw = dict(force_fp32=True) if arith == 'f16x3' else dict(sr_fp16=True)
    o = G.synthesis(ws, c, noise_mode='const', render_uniforms=(u1.to(DEV), u2.to(DEV)), **kw)
    g_img = O._randn('gf_img', 8, o['image'].shape) / (3 * 512 * 512)
    g_dep = O._randn('gf_dep', 8, o['image_depth'].shape) / (128 * 128)
    grads = torch.autograd.grad([o['image'], o['image_depth']], [ws, c] + leaves, [g_img.to(DEV), g_dep.to(DEV)])
    close(grads[0], d['dws'], 2e-3 if arith == 'f16x3' else 2e-2, 'full d ws (all weights trainable)')
    worst, bulk, bad = {}, {}, []
    for k, gv in zip(wkeys, grads[2:]):
        ref_norm, ref_max = [float(v) for v in d['wg_stat.' + k]]
        flat = gv.detach().flatten()
        assert torch.isfinite(flat).all(), k
        got = flat[t(d['wg_idx.' + k])].double().cpu()
        ref = torch.from_numpy(d['wg_val.' + k]).double()
        # one product of fp16-rounded operands in the SR head (64 % of the FLOPs): every upstream gradient passes through it
        # f16x3: the rendered feature image itself agrees with the reference's to ~1e-4 (fp32 softplus / transmittance products, importance
        # samples that land one texel over) and every gradient inherits that; observed 0.5e-4 .. 9e-4 of max|g|, varying from run to run
        # with the order of the atomically accumulated sums -- bound 3e-3
        tol = 3e-3 if arith == 'f16x3' else 5e-2
        # (round 4 held a third of that under the deterministic build.  Round 5 measured the four combinations {conv_v3 on / off} x {exact /
        #  atomic accumulation}: b128.conv1.bias sits at 2.4e-3 in three of them and at 3.7e-4 in one, b32.conv0.noise_const at 1.1e-3 / 5.5e-4 --
        #  the worst entries are lrelu-kink branches of a few pre-activations that flip with ANY one-ulp change upstream (another kernel's summation
        #  order as much as an atomic's), deterministic but not smaller in the exact build.  Same bound in both builds.)
        err = float((got - ref).abs().max())
        worst[k] = err / ref_max
        if got.numel() >= 64:           # (scalar parameters -- the noise strengths -- have one probe: their error is the maximum above)
            bulk[k] = float((got - ref).abs().median()) / ref_max          # the probe set's MEDIAN error: what an arithmetic fault would move (the maximum sits on kink flips)
        nrm = float(flat.double().norm())
        if err > tol * ref_max:
            bad.append(f'd {k}: probe err {err:.3e} > {tol} * max|g| {ref_max:.3e}')
        if abs(nrm - ref_norm) > 2 * tol * ref_norm:
            bad.append(f'd {k}: norm {nrm:.6e} vs {ref_norm:.6e}')
    print({k: f'{v:.1e}' for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]})
    print('median probe error / max|g|, worst tensors:', {k: f'{v:.1e}' for k, v in sorted(bulk.items(), key=lambda kv: -kv[1])[:6]})
    assert not bad, bad
    assert max(bulk.values()) <= (BULK_BOUND_F16X3 if arith == 'f16x3' else BULK_BOUND_F16X1), max(bulk.values())


def test_broadcast_latent_row_equals_repeat():
    """w-space projection (projectors/w_projector.py:117, ws = (w_opt + w_noise).repeat([1, num_ws, 1])): fused.broadcast_rows hands the style
    bank a stride-0 view of the one row (no repeat copy; the bank's backward delivers d w in row 0, no row sum) -- same styles bit for bit,
    d w equal to the repeat path's row sum up to the order of the atomically accumulated layer contributions."""
    from inv3d_amd import fused
    cfg, G = small_G()
    g = torch.Generator().manual_seed(9)
    w0 = torch.randn(1, 1, 32, generator=g).to(DEV)
    c = t(O.synth_cameras(1, seed=11))
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    uni = (u1.to(DEV), u2.to(DEV))
    gi = None
    res = []
    for mode in ('repeat', 'broadcast'):
        w = w0.clone().requires_grad_(True)
        ws = w.repeat(1, G.backbone.num_ws, 1) if mode == 'repeat' else fused.broadcast_rows(w, G.backbone.num_ws)
        o = G.synthesis(ws, c, noise_mode='const', render_uniforms=uni, force_fp32=True)
        if gi is None:
            gi = torch.randn(o['image'].shape, generator=g).to(DEV)
        o['image'].backward(gi)
        res.append((o['image'].detach(), o['image_raw'].detach(), w.grad.detach()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    close(res[1][2], res[0][2], 2e-5, 'd w through the broadcast row vs the repeat path')
    # a consumer that is not the style bank gets a correct [N,L,D] gradient too (row sum fall-back)
    w = w0.clone().requires_grad_(True)
    ws = fused.broadcast_rows(w, 5)
    (ws * torch.arange(5, device=DEV).view(1, 5, 1)).sum().backward()
    close(w.grad, torch.full_like(w0, 10.0), 1e-6, 'broadcast rows: generic consumer')


def test_cpu_tensors_fail_loudly():
    from inv3d_amd.torch_utils.ops import bias_act
    from inv3d_amd._lib import Eg3dHipError
    with pytest.raises(Eg3dHipError):
        bias_act.bias_act(torch.randn(4, 4), torch.randn(4))


def test_forward_sample_and_fma_entry_points(golden):
    """The remaining public entry points of the reference's generator API: G(z, c) = synthesis(mapping(z, c), c) (triplane.py:112-115),
    G.sample_mixed / G.sample (density queries, :92-110) and torch_utils.ops.fma (fma.py:17-60, un-broadcast gradients)."""
    from inv3d_amd.torch_utils.ops import fma
    d = golden('graph_small')
    cfg, G = small_G()
    c = t(d['c'])
    z = t(d['map_z'])
    u1, u2 = O.make_uniforms(cfg, 2, seed=4)
    uni = (u1.to(DEV), u2.to(DEV))
    with torch.no_grad():
        ws = G.mapping(z, c, truncation_psi=0.7, truncation_cutoff=5)
        a = G(z, c, truncation_psi=0.7, truncation_cutoff=5, noise_mode='const', render_uniforms=uni)
        b = G.synthesis(ws, c, noise_mode='const', render_uniforms=uni)
        close(a['image'], b['image'], 0, 'G(z,c)')
        pts = (torch.rand(2, 777, 3, device=DEV) - 0.5) * cfg.rendering['box_warp']
        dirs = torch.zeros_like(pts)
        s1 = G.sample_mixed(pts, dirs, ws, noise_mode='const')
        s2 = G.sample(pts, dirs, z, c, truncation_psi=0.7, truncation_cutoff=5, noise_mode='const')
        planes = G.backbone.synthesis(ws, noise_mode='const')
        s3 = G.renderer.run_model(planes.view(2, 3, 32, planes.shape[-2], planes.shape[-1]), G.decoder, pts, dirs, G.rendering_kwargs)
        close(s1['sigma'], s3['sigma'], 0, 'sample_mixed'); close(s2['rgb'], s3['rgb'], 0, 'sample')
        P = O.synth_params(cfg, 0)
        pl_ref = O.backbone_synthesis(P, cfg, ws.cpu(), noise_mode='const')
        rgb_ref, sig_ref = O.run_model(P, pl_ref.view(2, 3, 32, pl_ref.shape[-2], pl_ref.shape[-1]), pts.cpu(), cfg.rendering)
        close(s1['sigma'], sig_ref, 1e-4, 'sample_mixed vs oracle'); close(s1['rgb'], rgb_ref, 1e-4, 'sample_mixed rgb vs oracle')
    g = torch.Generator().manual_seed(0)
    av, bv, cv = torch.randn(2, 3, 4, generator=g), torch.randn(1, 3, 1, generator=g), torch.randn(4, generator=g)
    ar, br, cr = [v.clone().requires_grad_(True) for v in (av, bv, cv)]
    ag, bg, cg = [v.to(DEV).requires_grad_(True) for v in (av, bv, cv)]
    (ar * br + cr).square().sum().backward()
    fma.fma(ag, bg, cg).square().sum().backward()
    for x, y in ((ag, ar), (bg, br), (cg, cr)):
        assert x.grad.shape == y.grad.shape
        close(x.grad, y.grad, 1e-5, 'fma grad')


def test_two_live_graphs_of_one_generator_backward_together():
    """The cross-layer backward fusion keeps one producer record per forward: two forwards of the same layers whose graphs are alive
    at the same time (loss = f(G(ws1)) + f(G(ws2)), one backward) must give the sum of the two separate gradients."""
    from inv3d_amd import synthetic as S
    cfg = O.small_config()
    G = S.make_generator(w_dim=32, z_dim=32, plane_res=32, channel_base=256, channel_max=16, nrr=16, sr_in_res=16, sr_widths=(16, 8),
                         rendering_kwargs=cfg.rendering, device=DEV)
    S.load_synthetic_weights(G, 0)
    G.requires_grad_(False)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    kw = dict(noise_mode='const', force_fp32=True, render_uniforms=(u1.float().to(DEV), u2.float().to(DEV)))
    cam = O.synth_cameras(1, seed=2).float().to(DEV)
    wa = O.synth_ws(cfg, 1, seed=3).float().to(DEV).requires_grad_(True)
    wb = O.synth_ws(cfg, 1, seed=7).float().to(DEV).requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    pa, pb = torch.randn(1, 3, 64, 64, generator=g).to(DEV), torch.randn(1, 3, 64, 64, generator=g).to(DEV)
    ga, = torch.autograd.grad((G.synthesis(wa, cam, **kw)['image'] * pa).sum(), wa)
    gb, = torch.autograd.grad((G.synthesis(wb, cam, **kw)['image'] * pb).sum(), wb)
    la = (G.synthesis(wa, cam, **kw)['image'] * pa).sum()
    lb = (G.synthesis(wb, cam, **kw)['image'] * pb).sum()
    ja, jb = torch.autograd.grad(la + lb, [wa, wb])
    for got, ref, name in ((ja, ga, 'first'), (jb, gb, 'second')):
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), name


def test_rendering_resolution_other_than_the_sr_input():
    """SuperresolutionHybrid8XDC resizes its inputs when the neural-rendering resolution differs from its input resolution
    (superresolution.py:282-286: bilinear, antialiased): render at 24^2 into the 16^2 SR head, against the oracle, forward and d ws."""
    cfg, G = small_G()
    nrr = 24
    P = O.synth_params(cfg, 0)
    ws = O.synth_ws(cfg, 1, seed=3)
    cam = O.synth_cameras(1, seed=2)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4, nrr=nrr)
    wr = ws.clone().requires_grad_(True)
    ref = O.synthesis(P, cfg, wr, cam, u1, u2, noise_mode='const', nrr=nrr)
    pr = torch.randn(ref['image'].shape, generator=torch.Generator().manual_seed(1), dtype=ref['image'].dtype)
    gref, = torch.autograd.grad((ref['image'] * pr).sum(), wr)
    wg = ws.float().to(DEV).requires_grad_(True)
    out = G.synthesis(wg, cam.float().to(DEV), neural_rendering_resolution=nrr, noise_mode='const', force_fp32=True,
                      render_uniforms=(u1.float().to(DEV), u2.float().to(DEV)))
    G.neural_rendering_resolution = 16                                 # the setting is sticky, as in the reference (triplane.py:58-61)
    assert out['image'].shape == ref['image'].shape and out['image_raw'].shape[-1] == nrr
    close(out['image'], ref['image'], 1e-4, 'image at nrr 24')
    close(out['image_raw'], ref['image_raw'], 5e-5, 'image_raw at nrr 24')
    gg, = torch.autograd.grad((out['image'] * pr.float().to(DEV)).sum(), wg)
    close(gg, gref.float(), 2e-4, 'd ws at nrr 24')

def test_generator_with_the_128px_head():
    """A whole TriPlaneGenerator built around SuperresolutionHybrid2X (the reference's 128^2 configs: training/superresolution.py:94-122; neural
    rendering at 64^2): G.synthesis through the product's graph-replaying entry point, image / raw image against the oracle assembled from its
    pinned pieces (backbone + renderer, then the stand-alone 2X head), and the gradient into ws."""
    from inv3d_amd.training.triplane import TriPlaneGenerator
    cfg = O.small_config(nrr=64)
    rk = dict(cfg.rendering, superresolution_module='training.superresolution.SuperresolutionHybrid2X')
    G = TriPlaneGenerator(z_dim=cfg.z_dim, c_dim=25, w_dim=cfg.w_dim, img_resolution=128, img_channels=3, sr_num_fp16_res=4, mapping_kwargs={'num_layers': 2},
                          rendering_kwargs=rk, sr_kwargs={'fused_modconv_default': 'inference_only', 'w_dim': cfg.w_dim}, plane_resolution=cfg.plane_res,
                          channel_base=cfg.channel_base, channel_max=cfg.channel_max, fused_modconv_default='inference_only', conv_clamp=None).eval().float()
    G.neural_rendering_resolution = 64
    P = {k: v for k, v in O.synth_params(cfg, seed=0).items() if not k.startswith('superresolution.')}
    P.update(O.sr_head_params('2X', seed=3, w_dim=cfg.w_dim))
    missing, unexpected = G.load_state_dict(P, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    G = G.to(DEV)
    ws = O.synth_ws(cfg, 1, seed=1)
    c = O.synth_cameras(1, seed=2)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4, nrr=64)
    wr = ws.clone().requires_grad_(True)
    planes = O.backbone_synthesis(P, cfg, wr, noise_mode='const')
    ro, rd = O.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 64)
    feat, depth, _ = O.render(P, planes.view(1, 3, cfg.plane_channels // 3, planes.shape[-2], planes.shape[-1]), ro, rd, cfg.rendering, u1, u2)
    fimg = feat.permute(0, 2, 1).reshape(1, feat.shape[-1], 64, 64).contiguous()
    img_r = O.sr_head(P, '2X', fimg[:, :3], fimg, wr, sr_antialias=True, conv_clamp=256.0, noise_mode='none')
    g = O._randn('g2x', 1, img_r.shape) / img_r.numel() ** 0.5
    dws_r, = torch.autograd.grad(img_r, wr, g)
    wg = ws.to(DEV).requires_grad_(True)
    out = G.synthesis(wg, c.to(DEV), noise_mode='const', force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
    assert tuple(out['image'].shape) == (1, 3, 128, 128) and tuple(out['image_raw'].shape) == (1, 3, 64, 64)
    close(out['image'], img_r, 2e-4, '2X generator image')
    close(out['image_raw'], fimg[:, :3], 2e-4, '2X generator raw image')
    dws, = torch.autograd.grad(out['image'], wg, g.to(DEV))
    close(dws, dws_r, 2e-3, '2X generator d ws')


def test_per_layer_style_affines_equal_the_style_bank():
    """The style affines of every layer from ONE launch (fused.style_bank) vs the per-layer path (`styles = self.affine(w)` inside each layer: what
    runs when the bank does not apply), full-size generator, every deferral switch at its default: same image and same latent gradient.  On the
    per-layer path an affine's backward node runs BEFORE the previous block's toRGB node, so a split-K data gradient whose finish (and x.z
    style-gradient term) is deferred to that toRGB launch would hand the affine an incomplete `ds` -- the deferral therefore requires the bank
    (ADVICE r4, medium); this test fails if that condition goes."""
    from inv3d_amd import hipops as H, synthetic as S, fused
    G = S.make_generator(device=DEV)
    S.load_synthetic_weights(G)
    G.requires_grad_(False)
    G.graph_eager = False
    ws0 = S.synth_ws(G.backbone.num_ws, 512, 1).to(DEV)
    cam = S.synth_cameras(1).to(DEV)
    u1, u2 = S.make_uniforms(1, 128 * 128, 48, 48)
    res = {}
    keep = fused.style_bank
    calls = []
    try:
        for mode in ('bank', 'per_layer'):
            if mode == 'per_layer':
                fused.style_bank = lambda ws, entries: (calls.append(len(entries)), None)[1]
            ws = ws0.clone().requires_grad_(True)
            out = G.synthesis(ws, cam, noise_mode='const', force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
            out['image'].square().sum().backward()
            res[mode] = (out['image'].detach().clone(), ws.grad.clone())
    finally:
        fused.style_bank = keep
    assert calls, 'the per-layer pass never asked for the bank: the test did not exercise what it claims'
    scale = float(res['bank'][0].abs().max())
    assert float((res['bank'][0] - res['per_layer'][0]).abs().max()) <= 2e-5 * scale
    g0, g1 = res['bank'][1], res['per_layer'][1]
    assert float((g0 - g1).abs().max()) <= 1e-4 * float(g0.abs().max()), float((g0 - g1).abs().max()) / float(g0.abs().max())


@pytest.mark.parametrize('switch', ['DEFER_EPILOGUE', 'DEFER_DGRAD_FINISH'])
def test_deferred_finishing_passes_equal_the_separate_launches(switch):
    """hipops.DEFER_EPILOGUE: conv1's split-K finishing pass run inside the toRGB launch of the 4^2 .. 64^2 blocks (eg3d_torgb_small_params::pre_z;
    the layer output is written by that launch).  hipops.DEFER_DGRAD_FINISH: conv0's split-K data gradient finished (styles, style gradient) inside
    the previous block's toRGB backward launch (eg3d_torgb_small_bwd_params::add_scale).  Full-size generator (the split-K layers only exist
    there): image and latent gradient equal those of the separate eg3d_modconv_epilogue_fwd / eg3d_dgrad_finish launches up to the rounding of the
    atomically accumulated sums."""
    from inv3d_amd import hipops as H, synthetic as S
    G = S.make_generator(device=DEV)
    S.load_synthetic_weights(G)
    G.requires_grad_(False)
    G.graph_eager = False                      # per-launch path: the switch is read when the launches are issued
    ws0 = S.synth_ws(G.backbone.num_ws, 512, 1).to(DEV)
    cam = S.synth_cameras(1).to(DEV)
    u1, u2 = S.make_uniforms(1, 128 * 128, 48, 48)
    res = {}
    keep = getattr(H, switch)
    try:
        for mode in (False, True):
            setattr(H, switch, mode)
            ws = ws0.clone().requires_grad_(True)
            out = G.synthesis(ws, cam, noise_mode='const', force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
            out['image'].square().sum().backward()
            res[mode] = (out['image'].detach().clone(), ws.grad.clone())
    finally:
        setattr(H, switch, keep)
    scale = float(res[False][0].abs().max())
    assert float((res[False][0] - res[True][0]).abs().max()) <= 2e-5 * scale
    g0, g1 = res[False][1], res[True][1]
    assert float((g0 - g1).abs().max()) <= 1e-4 * float(g0.abs().max())
