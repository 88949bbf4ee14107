"""Oracle vs the committed golden vectors (generated from the imported reference by tests/golden/make_golden.py).
CPU only; this is the oracle's pin that travels to machines without /root/reference."""
import numpy as np
import pytest
import torch

from oracle import eg3d_oracle as O


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if a.numel() else 0.0
    assert err <= tol * scale, f'err {err:.3e} > {tol} * {scale:.3e}'


def test_bias_act(golden):
    d = golden('bias_act')
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        dim, clamp, gain, alpha = d[f'{k}_meta']
        x = t(d[f'{k}_x']).requires_grad_(True)
        b = t(d[f'{k}_b']).requires_grad_(True)
        y = O.bias_act(x, b, dim=int(dim), act=str(d[f'{k}_act']), alpha=None if alpha < 0 else float(alpha),
                       gain=None if gain < 0 else float(gain), clamp=None if clamp < 0 else float(clamp))
        close(y, t(d[f'{k}_y']), 1e-6)
        dx, db = torch.autograd.grad(y, [x, b], t(d[f'{k}_dy']))
        close(dx, t(d[f'{k}_dx']), 1e-6)
        close(db, t(d[f'{k}_db']), 1e-5)


def test_upfirdn2d(golden):
    d = golden('upfirdn2d')
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = d[f'{k}_meta']
        f = t(d[f'{k}_f'])
        f = None if f.numel() == 0 else f
        x = t(d[f'{k}_x']).requires_grad_(True)
        y = O.upfirdn2d(x, f, up=(int(m[0]), int(m[1])), down=(int(m[2]), int(m[3])), padding=[int(v) for v in m[4:8]],
                        flip_filter=bool(m[8]), gain=float(m[9]))
        close(y, t(d[f'{k}_y']), 1e-6)
        dx, = torch.autograd.grad(y, x, t(d[f'{k}_dy']))
        close(dx, t(d[f'{k}_dx']), 1e-6)
    x, f44 = t(d['w_x']), t(d['f44'])
    close(O.upsample2d(x, f44), t(d['w_upsample2d_y']), 1e-6)
    close(O.downsample2d(x, f44), t(d['w_downsample2d_y']), 1e-6)
    close(O.filter2d(x, f44), t(d['w_filter2d_y']), 1e-6)


def _flrelu_case(d, i):
    k = f'c{i}'
    m = d[f'{k}_meta']
    opt = lambda a: None if a.size == 0 else t(a)
    kw = dict(fu=opt(d[f'{k}_fu']), fd=opt(d[f'{k}_fd']), up=int(m[0]), down=int(m[1]), padding=[int(v) for v in m[2:6]], gain=float(m[6]),
              slope=float(m[7]), clamp=None if m[8] < 0 else float(m[8]), flip_filter=bool(m[9]))
    return k, kw, opt(d[f'{k}_b'])


def test_filtered_lrelu(golden):
    """oracle.filtered_lrelu vs the reference's _filtered_lrelu_ref outputs and gradients (fixtures from make_golden.py)."""
    d = golden('filtered_lrelu')
    for i in range(int(d['ncases'])):
        k, kw, b = _flrelu_case(d, i)
        x = t(d[f'{k}_x']).requires_grad_(True)
        if b is not None:
            b = b.requires_grad_(True)
        y = O.filtered_lrelu(x, b=b, **kw)
        close(y, t(d[f'{k}_y']), 1e-6)
        g = torch.autograd.grad(y, [x] + ([b] if b is not None else []), t(d[f'{k}_dy']))
        close(g[0], t(d[f'{k}_dx']), 1e-5)
        if b is not None:
            close(g[1], t(d[f'{k}_db']), 1e-5)


def test_conv2d_resample(golden):
    d = golden('conv2d_resample')
    f44 = t(d['f44'])
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        m = [int(v) for v in d[f'{k}_meta']]
        x = t(d[f'{k}_x']).requires_grad_(True)
        w = t(d[f'{k}_w']).requires_grad_(True)
        y = O.conv2d_resample(x, w, f=f44, up=m[0], down=m[1], padding=m[2:6], groups=m[6], flip_weight=bool(m[7]))
        close(y, t(d[f'{k}_y']), 1e-5)
        dx, dw = torch.autograd.grad(y, [x, w], t(d[f'{k}_dy']))
        close(dx, t(d[f'{k}_dx']), 1e-5)
        close(dw, t(d[f'{k}_dw']), 1e-5)


def test_modulated_conv2d(golden):
    d = golden('modulated_conv2d')
    f44 = t(d['f44'])
    for i in range(int(d['ncases'])):
        k = f'c{i}'
        up, demod, fused = [int(v) for v in d[f'{k}_meta']]
        x = t(d[f'{k}_x']).requires_grad_(True)
        w = t(d[f'{k}_w']).requires_grad_(True)
        s = t(d[f'{k}_s']).requires_grad_(True)
        nz = t(d[f'{k}_noise'])
        nz = None if nz.numel() == 0 else nz
        y = O.modulated_conv2d(x, w, s, noise=nz, up=up, padding=1, resample_filter=f44, demodulate=bool(demod),
                               flip_weight=(up == 1), fused_modconv=bool(fused))
        close(y, t(d[f'{k}_y']), 1e-5)
        dx, dw, ds = torch.autograd.grad(y, [x, w, s], t(d[f'{k}_dy']))
        close(dx, t(d[f'{k}_dx']), 1e-5)
        close(dw, t(d[f'{k}_dw']), 1e-5)
        close(ds, t(d[f'{k}_ds']), 1e-5)


def test_fully_connected(golden):
    d = golden('fully_connected')
    for i in range(int(d['ncases'])):
        y = O.fully_connected(t(d[f'c{i}_x']), t(d[f'c{i}_w']), t(d[f'c{i}_b']), float(d[f'c{i}_lr']), str(d[f'c{i}_act']))
        close(y, t(d[f'c{i}_y']), 1e-6)


def test_renderer_pieces(golden):
    d = golden('renderer')
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=5)
    opts = dict(cfg.rendering)
    o, dr = O.ray_sampler(t(d['rs_c2w']), t(d['rs_K']), 8)
    close(o, t(d['rs_o']), 1e-6); close(dr, t(d['rs_d']), 1e-6)
    close(O.sample_from_planes(t(d['sp_planes']), t(d['sp_coords']), 1.0), t(d['sp_feats']), 1e-6)
    rgb, sig = O.osg_decoder(P, t(d['dec_in']))
    close(rgb, t(d['dec_rgb']), 1e-6); close(sig, t(d['dec_sigma']), 1e-6)
    for wb in (0, 1):
        r, dp, w = O.ray_march(t(d['rm_colors']), t(d['rm_dens']), t(d['rm_depths']), dict(opts, white_back=bool(wb)))
        close(r, t(d[f'rm{wb}_rgb']), 1e-6); close(dp, t(d[f'rm{wb}_depth']), 1e-6); close(w, t(d[f'rm{wb}_w']), 1e-6)
    u1 = t(d['ss_u1'])
    close(O.sample_stratified(2, 64, 2.25, 3.3, 12, False, u1), t(d['ss_fixed']), 1e-6)
    close(O.sample_stratified(2, 64, 2.25, 3.3, 12, True, u1), t(d['ss_disp']), 1e-6)
    close(O.sample_stratified(2, 64, t(d['ss_rs']), t(d['ss_re']), 12, False, u1), t(d['ss_tensor']), 1e-6)
    close(O.sample_importance(t(d['si_z']), t(d['si_w']), 12, t(d['si_u2'])), t(d['si_out']), 1e-6)
    ud, uc, us = O.unify_samples(*[t(d[k]) for k in ('us_d1', 'us_c1', 'us_s1', 'us_d2', 'us_c2', 'us_s2')])
    close(ud, t(d['us_d']), 0)
    # torch.sort in the reference is not stable: at exact depth ties the order (hence colours) is undefined -> mask ties
    dd = ud[..., 0]
    tie = torch.zeros_like(dd, dtype=torch.bool)
    tie[..., 1:] |= dd[..., 1:] == dd[..., :-1]
    tie[..., :-1] |= dd[..., 1:] == dd[..., :-1]
    assert int(tie.sum()) == 2
    close(uc[~tie], t(d['us_c'])[~tie], 0); close(us[~tie], t(d['us_s'])[~tie], 0)
    q0, q1 = O.get_ray_limits_box(t(d['box_o']), t(d['box_d']), 1.0)
    close(q0, t(d['box_tmin']), 1e-6); close(q1, t(d['box_tmax']), 1e-6)


def test_render_forward_backward(golden):
    d = golden('renderer')
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=5)
    planes = t(d['rn_planes']).requires_grad_(True)
    c2w = t(d['rn_c2w']).requires_grad_(True)
    o, dr = O.ray_sampler(c2w, t(d['rn_K']), 6)
    rgb, dep, ws = O.render(P, planes, o, dr, cfg.rendering, t(d['rn_u1']), t(d['rn_u2']))
    close(rgb, t(d['rn_rgb']), 2e-6); close(dep, t(d['rn_depth']), 2e-6); close(ws, t(d['rn_wsum']), 2e-6)
    g = torch.autograd.grad([rgb, dep], [planes, c2w], [t(d['rn_grgb']), t(d['rn_gdepth'])])
    close(g[0], t(d['rn_dplanes']), 1e-5); close(g[1], t(d['rn_dc2w']), 1e-5)
    o_auto = dict(cfg.rendering, ray_start='auto', ray_end='auto')
    rgb, dep, _ = O.render(P, planes, o, dr, o_auto, t(d['rn_u1']), t(d['rn_u2']))
    close(rgb, t(d['rn_auto_rgb']), 2e-6); close(dep, t(d['rn_auto_depth']), 2e-6)


@pytest.mark.parametrize('mode', ['const', 'random'])
def test_graph_small(golden, mode):
    d = golden('graph_small')
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=0)
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
                noises[nm] = O._randn('noise.' + nm, 6, (n, 1, r, r))
    o = O.synthesis(P, cfg, ws, c, u1, u2, noise_mode=mode, noises=noises)
    m = mode[0]
    close(o['image'], t(d[f'{m}_image']), 2e-5)
    close(o['image_raw'], t(d[f'{m}_raw']), 2e-5)
    close(o['image_depth'], t(d[f'{m}_depth']), 2e-5)
    close(o['planes'], t(d[f'{m}_planes']), 2e-5)
    g = torch.autograd.grad([o['image'], o['image_raw'], o['image_depth']], [ws, c],
                            [t(d['g_img']), t(d['g_raw']), t(d['g_dep'])])
    close(g[0], t(d[f'{m}_dws']), 2e-4); close(g[1], t(d[f'{m}_dc']), 2e-4)


def test_mapping_and_glue(golden):
    d = golden('graph_small')
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=0)
    close(O.mapping(P, cfg, t(d['map_z']), t(d['c']), 0.7, 5), t(d['map_out']), 1e-5)
    g = golden('loss_glue')
    close(O.quaternion_to_rotmat(t(g['q'])), t(g['R']), 1e-6)
    close(O.compute_tv_norm(t(g['tv_in'])), t(g['tv']), 1e-7)
    close(O.noise_regularizer([t(g[f'reg_buf{i}']) for i in range(4)]), t(g['reg']), 1e-6)
    close(O.lookat_cam2world(t(g['lookat_origin'])[0], torch.zeros(3)), t(g['lookat'])[0], 1e-6)


def test_param_schema_counts():
    """132 parameter tensors + 44 buffers, 30.66 M params at the full config."""
    cfg = O.full_config()
    sh = O.param_shapes(cfg)
    bufs = [k for k in sh if k.endswith(O.BUFFER_SUFFIXES)]
    params = [k for k in sh if k not in bufs]
    assert len(params) == 132 and len(bufs) == 44
    total = sum(int(np.prod(sh[k])) for k in params)
    assert abs(total - 30.66e6) < 0.02e6, total
    assert cfg.num_ws == 14


def test_inference_consumers(golden):
    """oracle/inference_oracle.py and the host-side camera / grid helpers of inv3d_amd/inference.py vs the reference's
    LookAtPoseSampler, create_samples, create_geometry evaluation and mean-latent statistics (fixture `inference`)."""
    from oracle import inference_oracle as IO
    from inv3d_amd import inference as INF
    d = golden('inference')
    for (h, v), m in zip(d['hv'], d['poses']):
        close(IO.lookat_pose(float(h), float(v), (0., 0., 0.), 2.7), t(m), 1e-6)
        close(INF.lookat_pose(float(h), float(v), (0., 0., 0.), 2.7), t(m), 1e-6)
    close(IO.orbit_cameras(8), t(d['orbit8']), 1e-6)
    close(INF.orbit_cameras(8), t(d['orbit8']), 1e-6)
    close(IO.create_samples(20, 1.0), t(d['samples20']), 1e-6)
    close(INF._grid_points(20, 1.0, 0, 8000, 'cpu').unsqueeze(0), t(d['samples20']), 1e-6)
    close(INF._grid_points(20, 1.0, 777, 4001, 'cpu'), t(d['samples20'])[0, 777:4001], 1e-6)
    cfg = O.small_config()
    P = O.synth_params(cfg, seed=0)
    close(IO.density_grid(P, cfg, t(d['grid_ws']), 12), t(d['grid12']), 2e-5)
    w_avg, w_std = IO.w_stats(P, cfg, 64)
    close(w_avg, t(d['w_avg64']), 1e-5)
    assert abs(w_std - float(d['w_std64'])) <= 1e-5 * max(1.0, float(d['w_std64']))


def test_pose_net_oracle(golden):
    """oracle/pose_net_oracle.py vs the reference's ResNet-34 pose estimator (outputs and parameter gradients, fixture `pose_net`)."""
    from oracle import pose_net_oracle as PO
    d = golden('pose_net')
    keys = [k[len('d4_g.'):] for k in d.files if k.startswith('d4_g.')] + [k[len('d4_gs.'):] for k in d.files if k.startswith('d4_gs.')]
    for dims in (4, 6):
        sd = PO.synth_state(seed=3, output_dims=dims)
        sd = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
        y = PO.forward(sd, t(d[f'd{dims}_img']))
        close(y, t(d[f'd{dims}_y']), 1e-6)
        grads = torch.autograd.grad(y, [sd[k] for k in keys], t(d[f'd{dims}_gy']))
        for k, g in zip(keys, grads):
            if f'd{dims}_g.{k}' in d.files:
                close(g, t(d[f'd{dims}_g.{k}']), 1e-5)
            else:
                close(g.flatten()[::97], t(d[f'd{dims}_gs.{k}']), 1e-5)
