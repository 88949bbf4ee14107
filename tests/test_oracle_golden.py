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


def test_loss_glue(golden):
    """Rotation parametrisations, pose -> camera block, noise regulariser, depth TV, line-plane intersection: fixture `loss_glue` holds the
    outputs of the reference's own functions / lifted statements (utils/camera_utils.py, w_projector.py:147-172,221-237,
    base_coach.py:294-305, warping_loss.py:58-72).  Checked: the oracle AND the product's host-side restatements (pure torch, CPU)."""
    from oracle import inversion_oracle as IO
    from inv3d_amd import inversion as INV
    g = golden('loss_glue')
    for mod in (O, INV):
        close(mod.quaternion_to_rotmat(t(g['q'])), t(g['R']), 1e-6)
        close(mod.rot6d_to_rotmat(t(g['x6'])), t(g['R6']), 1e-6)
        close(torch.cat([mod.pose_to_rotmat(a[None], 'euler') for a in t(g['ang'])]), t(g['Re']), 1e-6)
        close(mod.euler_to_rotmat(torch.tensor([1.2]), torch.tensor([1.9]), torch.tensor([[0.3]])), t(g['roll_R']), 1e-6)
        close(mod.compute_tv_norm(t(g['tv_in'])), t(g['tv']), 1e-7)
    close(O.noise_regularizer([t(g[f'reg_buf{i}']) for i in range(6)]), t(g['reg']), 1e-6)
    close(O.lookat_cam2world(t(g['lookat_origin'])[0], torch.zeros(3)), t(g['lookat'])[0], 1e-6)
    intr = torch.tensor([4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]).unsqueeze(0)
    for mode in ('quat', '6d', 'euler'):
        for mod, p2c in ((O, IO.pose_to_cam), (INV, INV.pose_to_cam)):
            pr, tr = t(g[f'pose_{mode}_pred']).requires_grad_(True), t(g[f'pose_{mode}_tr']).requires_grad_(True)
            _, cam = p2c(mod.pose_to_rotmat(pr, mode), tr, intr, 2.7)
            close(cam, t(g[f'pose_{mode}_cam']), 1e-6)
            d_pr, d_tr = torch.autograd.grad(cam, [pr, tr], t(g[f'pose_{mode}_gcam']))
            close(d_pr, t(g[f'pose_{mode}_dpred']), 2e-5)
            close(d_tr, t(g[f'pose_{mode}_dtr']), 2e-5)
    args = [t(g[k]) for k in ('lpc_n', 'lpc_p', 'lpc_d', 'lpc_o')]
    close(IO.line_plane_collision(*args), t(g['lpc_out']), 1e-5)
    close(INV.line_plane_intersection(*args), t(g['lpc_out']), 1e-5)


def adam_close(a, b, tol, step_bound, frac=0.995):
    """State after a few Adam steps (see make_golden.check_adam): an element whose gradient is rounding noise takes a full step in an
    undetermined direction, so >= frac of the elements must agree to tol and every element to the accumulated step size."""
    err = (torch.as_tensor(a) - torch.as_tensor(b)).abs()
    scale = max(1.0, float(torch.as_tensor(b).abs().max()))
    assert float((err <= tol * scale).float().mean()) >= frac and float(err.max()) <= step_bound, float(err.max())


@pytest.mark.parametrize('mode', ['quat', '6d', 'euler'])
def test_projector_loop_pin(golden, mode):
    """ProjectorOracle vs the reference's own Phase-A loop body (w_projector.py:145-270, lifted and run by make_golden.py with the
    reference's calc_warping_loss / RaySampler / generator classes): loss, feature distance, regulariser and warping loss of every
    step (2 camera-preheat + 4 full steps), final latent, pose-estimator parameters, translation and noise buffers."""
    from oracle import inversion_oracle as IO
    d = golden('projector_loop')
    cfg = IO.pin_config()
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)
    close(target.flatten()[::37], t(d['target_probe']), 1e-6)
    pin = IO.pin_projector_inputs(cfg, P, mode)
    lr = dict(quat=6e-7, euler=6e-6)
    lr['6d'] = 6e-6                                                        # configs/hyperparameters.py cam_lr_{quat,2d,6d}
    po = IO.ProjectorOracle(P, cfg, target[None], num_steps=IO.PIN_PROJ_STEPS, optimize_pose=True, use_warping_loss=True, init_noise=pin['init_noise'],
                            w_start=pin['w0'], cam_preheat_steps=IO.PIN_PROJ_PREHEAT, pose_mode=mode, pose_net=IO.StubPoseNet(pin['pose_base'], seed=7),
                            w_std=IO.PIN_W_STD, translation_start=IO.PIN_TRANSLATION_START, cam_lr=lr[mode])
    trace = []
    for k in range(IO.PIN_PROJ_STEPS):
        r = po.step(*pin['uniforms'][k], w_noise=pin['wns'][k])
        trace.append((float(r['loss']), float(r['dist']), float(r['reg']), float(r['warp']), float(O.psnr_01(r['image'], target[None]))))
    ref = t(d[f'{mode}_trace'])
    assert abs(trace[-1][4] - float(ref[-1, 4])) <= 1e-3              # final-PSNR drift <= 1e-3 dB
    for j in range(5):
        close(torch.tensor(trace)[:, j], ref[:, j].float(), 2e-5)
    close(po.w_opt.detach(), t(d[f'{mode}_w_opt']), 1e-5)
    close(po.translation_opt.detach(), t(d[f'{mode}_translation']), 1e-5)
    close(po.pose_net.base.detach(), t(d[f'{mode}_pose_base']), 1e-6)
    close(po.pose_net.A.detach(), t(d[f'{mode}_pose_A']), 1e-6)
    adam_close(po.bufs[-1].detach(), t(d[f'{mode}_buf_last']), 1e-5, IO.PIN_PROJ_STEPS * 0.01)
    adam_close(po.bufs2[-1].detach(), t(d[f'{mode}_srbuf_last']), 1e-5, IO.PIN_PROJ_STEPS * 0.01)


def test_c3_full_pin_first_step(golden):
    """Config C3 at full size: the oracle's camera-preheat step against the reference's loop body on the full-size generator
    (make_golden.py::gen_c3_full checks both recorded steps; here the first, to keep the CPU suite short)."""
    from oracle import inversion_oracle as IO
    d = golden('c3_full')
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)
    close(target.flatten()[::4099], t(d['target_probe']), 1e-6)
    pin = IO.pin_projector_inputs(cfg, P, 'quat')
    po = IO.ProjectorOracle(P, cfg, target[None], num_steps=2, optimize_pose=True, use_warping_loss=True, init_noise=pin['init_noise'],
                            w_start=pin['w0'], cam_preheat_steps=1, pose_mode='quat', pose_net=IO.StubPoseNet(pin['pose_base'], seed=7),
                            w_std=IO.PIN_W_STD, translation_start=IO.PIN_TRANSLATION_START, cam_lr=6e-7)
    r = po.step(*pin['uniforms'][0], w_noise=pin['wns'][0])
    got = torch.tensor([float(r['loss']), float(r['dist']), float(r['reg']), float(r['warp']), float(O.psnr_01(r['image'], target[None]))])
    close(got, t(d['trace'])[0].float(), 5e-5)
    close(po.translation_opt.grad, t(d['d_translation'])[0], 1e-4)


def test_tuner_loop_pin(golden):
    """PivotalTunerOracle vs the reference's own Phase-B loop (single_id_coach.py:64-77 with BaseCoach.calc_loss / forward and
    compute_tv_norm, lifted): per-step loss / MSE / LPIPS-stub, tuned weights, and the LPIPS-threshold exit before the update."""
    from oracle import inversion_oracle as IO
    d = golden('tuner_loop')
    cfg = IO.pin_config(tuner=True)
    P = O.synth_params(cfg, seed=0)
    target = IO.pin_target(cfg, P)[None]
    close(target.flatten()[::997], t(d['target_probe']), 1e-6)
    pin = IO.pin_tuner_inputs(cfg)
    for tag in ('full', 'stop'):
        to = IO.PivotalTunerOracle(P, cfg, target, pin['w_pivot'], pin['cam'], lr=3e-4, lpips_threshold=float(d[f'{tag}_thr']))
        trace = []
        for k in range(IO.PIN_TUNER_STEPS):
            r = to.step(*pin['uniforms'][k], noise_mode='random', noises=pin['noises'][k], early_stop=True)
            if r['done']:
                break
            trace.append((float(r['loss']), float(r['l2']), float(r['lpips']), float(O.psnr_01(r['image'], target))))
        ref = t(d[f'{tag}_trace']).float()
        assert len(trace) == ref.shape[0] == (5 if tag == 'full' else 3)
        close(torch.tensor(trace), ref, 2e-5)
        for key in [k[len(tag) + 3:] for k in d.files if k.startswith(tag + '_p.')]:
            close(to.P[key].detach(), t(d[f'{tag}_p.{key}']), 1e-5)
        if tag == 'stop':
            break


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


def test_e4e_oracle(golden):
    """oracle/e4e_oracle.py vs the reference's Encoder4Editing(50, 'ir_se') (fixture `e4e`: [0,255] input, all 18 codes)."""
    from oracle import e4e_oracle as EO
    d = golden('e4e')
    sd = EO.synth_state(seed=5)
    with torch.no_grad():
        y = EO.forward(sd, t(d['img']))
    close(y, t(d['codes']), 2e-5)


SR_KINDS = ('8X', '4X', '2X', 'Deepfp32')
SR_LEAVES = ['block0.conv0.weight', 'block0.conv1.noise_strength', 'block1.conv0.weight', 'block1.torgb.weight', 'block1.torgb.bias']


def _sr_inputs(kind, tag):
    in_res = O.SR_HEADS[kind][0]
    r = in_res if tag == 'own' else in_res // 2
    x = O._randn(f'srx.{kind}.{tag}', 5, (1, 32, r, r))
    rgb = O._randn(f'srrgb.{kind}.{tag}', 5, (1, 3, r, r))
    ws = O._randn(f'srws.{kind}', 5, (1, 14, 512))
    out_res = O.SR_HEADS[kind][3]
    g = O._randn(f'srg.{kind}', 5, (1, 3, out_res, out_res)) / (3 * out_res * out_res) ** 0.5
    return x, rgb, ws, g


@pytest.mark.parametrize('kind', SR_KINDS)
def test_sr_heads_pin(golden, kind):
    """The reference's other super-resolution heads (training/superresolution.py:29-152; 4X / 2X / Deepfp32 start with a SynthesisBlockNoUp,
    :155-262): the oracle against probes recorded from the reference classes, image and every gradient, at the head's own input size and
    through its bilinear resize."""
    d = golden('sr_heads')
    P = O.sr_head_params(kind, seed=3)
    for tag in ('own', 'small'):
        x, rgb, ws, g = _sr_inputs(kind, tag)
        Pg = {k: v.clone().requires_grad_(k[len('superresolution.'):] in SR_LEAVES) for k, v in P.items()}
        x, rgb, ws = x.requires_grad_(True), rgb.requires_grad_(True), ws.requires_grad_(True)
        img = O.sr_head(Pg, kind, rgb, x, ws, sr_antialias=True, conv_clamp=256.0, noise_mode='const')
        close(img.flatten()[torch.from_numpy(d[f'{kind}.{tag}.idx'])], t(d[f'{kind}.{tag}.img']), 1e-5)
        grads = torch.autograd.grad(img, [x, rgb, ws] + [Pg['superresolution.' + n] for n in SR_LEAVES], g)
        for nm, gv in zip(['x', 'rgb', 'ws'] + SR_LEAVES, grads):
            close(gv.flatten()[torch.from_numpy(d[f'{kind}.{tag}.gidx.{nm}'])], t(d[f'{kind}.{tag}.gval.{nm}']), 2e-5)
