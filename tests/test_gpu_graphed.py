"""The drop-in caller's path: a PLAIN loop over G.synthesis (the shape of training/projectors/w_projector.py:189-261 and of
training/coaches/single_id_coach.py:64-77 -- no projector / tuner object of this package involved) must (i) be replayed from HIP graphs
once its call signature repeats (inv3d_amd/graphed.py), (ii) give the results of the per-launch path, (iii) not be host-bound at full size."""
import time

import pytest
import torch
import torch.nn.functional as F

from oracle import eg3d_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _small():
    from inv3d_amd import synthetic as S
    cfg = O.small_config()
    G = S.make_generator(w_dim=32, z_dim=32, plane_res=32, channel_base=256, channel_max=16, nrr=16, sr_in_res=16, sr_widths=(16, 8),
                         rendering_kwargs=cfg.rendering, device=DEV)
    S.load_synthetic_weights(G, 0)
    return cfg, G


def _phase_a_loop(G, cfg, steps, graph_eager, lr=0.01):
    """w_projector.py's step, plainly: latent + backbone noise maps as leaves, Adam, loss on the image, backward, update."""
    G.graph_eager = graph_eager
    G.requires_grad_(False)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    uni = (u1.to(DEV), u2.to(DEV))
    cam = O.synth_cameras(1, seed=2).to(DEV)
    target = O._randn('tgt', 5, (1, 3, 64, 64)).to(DEV).clamp(-1, 1)
    w_opt = O.synth_ws(cfg, 1, seed=1)[:, :1].to(DEV).clone().requires_grad_(True)
    bufs = [b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n]
    for b in bufs:
        b.requires_grad = True
    opt = torch.optim.Adam([w_opt] + bufs, lr=lr)
    losses = []
    for i in range(steps):
        ws = (w_opt + 0.01 * O._randn('wn', i, (1, 1, cfg.w_dim)).to(DEV)).repeat(1, cfg.num_ws, 1)
        out = G.synthesis(ws, cam, noise_mode='const', force_fp32=True, render_uniforms=uni)
        loss = (out['image'] - target).square().mean() + 0.1 * out['image_depth'].mean() + 0.01 * out['image_raw'].abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    res = (w_opt.detach().clone(), [b.detach().clone() for b in bufs], losses, out['image'].detach().clone())
    for b in bufs:
        b.requires_grad = False
    return res


def test_plain_phase_a_loop_replays_and_matches_the_per_launch_path():
    from inv3d_amd import graphed
    cfg, G1 = _small()
    _, G2 = _small()
    before = dict(graphed.STATS)
    a = _phase_a_loop(G1, cfg, 9, True)
    assert graphed.STATS['captured'] == before['captured'] + 1, graphed.STATS
    assert graphed.STATS['replayed'] >= before['replayed'] + 7 - 1, graphed.STATS
    b = _phase_a_loop(G2, cfg, 9, False)
    assert float((a[0] - b[0]).abs().max()) < 1e-5
    for x, y in zip(a[1], b[1]):
        assert float((x - y).abs().max()) < 1e-4
    assert max(abs(p - q) for p, q in zip(a[2], b[2])) < 1e-5 * max(1.0, abs(b[2][0]))
    assert float((a[3] - b[3]).abs().max()) < 1e-4


@pytest.mark.parametrize('set_to_none', [True, False])
def test_plain_phase_b_loop_with_trainable_weights(set_to_none):
    """single_id_coach.py:64-77 plainly: every generator weight trainable, torch.optim.Adam, noise_mode='const' so both runs see the same
    noise.  The captured forward re-packs the weight images itself, so optimiser steps between replays are seen; zero_grad(set_to_none=
    False) leaves .grad aliasing a static buffer, which the node must not accumulate onto itself."""
    cfg, Ga = _small()
    _, Gb = _small()
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    uni = (u1.to(DEV), u2.to(DEV))
    cam = O.synth_cameras(1, seed=2).to(DEV)
    ws = O.synth_ws(cfg, 1, seed=1).to(DEV)
    target = O._randn('tgt', 5, (1, 3, 64, 64)).to(DEV).clamp(-1, 1)
    outs = []
    for G, flag in ((Ga, True), (Gb, False)):
        G.graph_eager = flag
        G.requires_grad_(True)
        opt = torch.optim.Adam(G.parameters(), lr=3e-4)
        ls = []
        for i in range(8):
            gen = G.synthesis(ws, cam, noise_mode='const', render_uniforms=uni)
            loss = F.mse_loss(gen['image'], target) + F.mse_loss(gen['image_raw'], F.avg_pool2d(target, 4)) + gen['image_depth'].square().mean()
            opt.zero_grad(set_to_none=set_to_none)
            loss.backward()
            opt.step()
            ls.append(float(loss))
        outs.append((ls, {k: v.detach().clone() for k, v in G.named_parameters()}))
    la, lb = outs[0][0], outs[1][0]
    assert max(abs(p - q) for p, q in zip(la, lb)) <= 2e-5 * max(1.0, abs(lb[0])), (la, lb)
    assert lb[-1] < lb[0]
    for k, v in outs[1][1].items():
        d = float((outs[0][1][k] - v).abs().max())
        assert d <= 8 * 3e-4 * 0.02 + 1e-6, (k, d)          # Adam: elements whose gradient is at rounding level may differ by a fraction of the step


def test_no_grad_calls_replay_and_match():
    """Orbit frames / evaluation renders: repeated no-grad calls with a changing camera."""
    from inv3d_amd import graphed
    cfg, Ga = _small()
    _, Gb = _small()
    Ga.graph_eager, Gb.graph_eager = True, False
    ws = O.synth_ws(cfg, 1, seed=1).to(DEV)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    uni = (u1.to(DEV), u2.to(DEV))
    cams = O.synth_cameras(6, seed=2).to(DEV)
    n0 = graphed.STATS['replayed']
    with torch.no_grad():
        for i in range(6):
            a = Ga.synthesis(ws, cams[i:i + 1], noise_mode='const', render_uniforms=uni)
            b = Gb.synthesis(ws, cams[i:i + 1], noise_mode='const', render_uniforms=uni)
            for k in ('image', 'image_raw', 'image_depth'):
                assert float((a[k] - b[k]).abs().max()) <= 1e-5, (i, k)
    assert graphed.STATS['replayed'] >= n0 + 3


def test_no_grad_preview_between_optimiser_steps_sees_the_new_weights():
    """single_id_coach's pattern: a no-grad preview render (hot after three calls -> captured), then optimiser steps on the
    generator's weights, then the same preview call again.  The replay must render with the UPDATED weights: trainable weights are
    re-packed inside the captured forward whether or not the call differentiates."""
    from inv3d_amd import graphed
    cfg, Ga = _small()
    _, Gb = _small()
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    uni = (u1.to(DEV), u2.to(DEV))
    cam = O.synth_cameras(1, seed=2).to(DEV)
    ws = O.synth_ws(cfg, 1, seed=1).to(DEV)
    target = O._randn('tgt', 5, (1, 3, 64, 64)).to(DEV).clamp(-1, 1)
    res = []
    for G, flag in ((Ga, True), (Gb, False)):
        G.graph_eager = flag
        G.requires_grad_(True)
        opt = torch.optim.Adam(G.parameters(), lr=3e-3)
        n0 = graphed.STATS['replayed']
        with torch.no_grad():
            for _ in range(5):
                before = G.synthesis(ws, cam, noise_mode='const', render_uniforms=uni)['image'].clone()
        if flag:
            assert graphed.STATS['replayed'] >= n0 + 2
        for i in range(4):
            gen = G.synthesis(ws, cam, noise_mode='const', render_uniforms=uni)
            loss = F.mse_loss(gen['image'], target)
            opt.zero_grad()
            loss.backward()
            opt.step()
        with torch.no_grad():
            after = G.synthesis(ws, cam, noise_mode='const', render_uniforms=uni)['image'].clone()
        res.append((before, after))
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-5
    moved = float((res[1][1] - res[1][0]).abs().max())
    assert moved > 1e-3, 'the optimiser steps must change the render for this test to mean anything'
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-3 * max(1.0, moved), 'replayed preview rendered with stale weights'


def test_rendering_kwargs_are_part_of_the_signature():
    """viz/renderer.py and gen_videos.py rewrite G.rendering_kwargs between calls (depth resolutions, ray range): a captured graph must not
    be replayed for other values."""
    cfg, Ga = _small()
    _, Gb = _small()
    Ga.graph_eager, Gb.graph_eager = True, False
    ws = O.synth_ws(cfg, 1, seed=1).to(DEV)
    cam = O.synth_cameras(1, seed=2).to(DEV)
    with torch.no_grad():
        for _ in range(4):
            Ga.synthesis(ws, cam, noise_mode='const')
        for G in (Ga, Gb):
            G.rendering_kwargs['ray_start'] = float(G.rendering_kwargs['ray_start']) + 0.15
            G.rendering_kwargs['white_back'] = True
        torch.manual_seed(5)
        a = Ga.synthesis(ws, cam, noise_mode='const')
        torch.manual_seed(5)
        b = Gb.synthesis(ws, cam, noise_mode='const')
    for k in ('image_raw', 'image_depth'):
        assert float((a[k] - b[k]).abs().max()) <= 1e-4, k


def test_second_forward_before_the_first_backward_falls_back():
    """Two live graphs of one signature: the second forward must not overwrite the first one's captured activations."""
    from inv3d_amd import graphed
    cfg, G = _small()
    G.requires_grad_(False)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    kw = dict(noise_mode='const', force_fp32=True, render_uniforms=(u1.to(DEV), u2.to(DEV)))
    cam = O.synth_cameras(1, seed=2).to(DEV)
    ws = [O.synth_ws(cfg, 1, seed=s).to(DEV).requires_grad_(True) for s in (3, 7)]
    p = O._randn('p', 1, (1, 3, 64, 64)).to(DEV)
    sep = []
    G.graph_eager = False
    for w in ws:
        sep.append(torch.autograd.grad((G.synthesis(w, cam, **kw)['image'] * p).sum(), w)[0])
    G.graph_eager = True
    for _ in range(3):          # make the signature hot (captured on the third call)
        torch.autograd.grad((G.synthesis(ws[0], cam, **kw)['image'] * p).sum(), ws[0])
    n_fb = graphed.STATS['fallback_pending']
    la = (G.synthesis(ws[0], cam, **kw)['image'] * p).sum()
    lb = (G.synthesis(ws[1], cam, **kw)['image'] * p).sum()
    assert graphed.STATS['fallback_pending'] == n_fb + 1
    ga, gb = torch.autograd.grad(la + lb, ws)
    for got, ref in ((ga, sep[0]), (gb, sep[1])):
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_full_size_plain_loop_is_not_host_bound():
    """`for step: out = G.synthesis(ws, c, noise_mode='const', force_fp32=True); loss.backward(); opt.step()` on the
    ffhqrebalanced512-128-shaped generator (the shape of w_projector.py:189-261) at <= 8 ms per step -- the per-launch path needs ~16 ms
    of host time for the same ~6 ms of GPU work -- and the same trajectory as the per-launch path."""
    from inv3d_amd import synthetic as S, graphed
    G = S.make_generator(device=DEV)
    S.load_synthetic_weights(G, 0)
    G.requires_grad_(False)
    cam = S.synth_cameras(1, seed=2).to(DEV)
    with torch.no_grad():
        target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(DEV), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
        t256 = F.avg_pool2d(target, 2)
    bufs = [b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n]
    for b in bufs:
        b.requires_grad = True

    cfg = O.full_config()
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    pinned = (u1.to(DEV), u2.to(DEV))

    def run(flag, steps, timed_from, uni=None):
        G.graph_eager = flag
        gen = torch.Generator(device=DEV).manual_seed(0)
        w_opt = S.synth_ws(14, 512, 1, seed=1)[:, :1].to(DEV).clone().requires_grad_(True)
        opt = torch.optim.Adam([w_opt] + bufs, lr=0.01, fused=True)
        t0 = None
        for i in range(steps):
            if i == timed_from:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ws = (w_opt + 0.01 * torch.randn(w_opt.shape, device=DEV, generator=gen)).repeat(1, 14, 1)
            out = G.synthesis(ws, cam, noise_mode='const', force_fp32=True, render_uniforms=uni)
            loss = (F.avg_pool2d(out['image'], 2) - t256).square().sum()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (steps - timed_from) * 1e3, w_opt.detach().clone(), float(loss)

    saved = [b.detach().clone() for b in bufs]
    n0 = graphed.STATS['replayed']
    ms, w_a, l_a = run(True, 24, 4)
    assert graphed.STATS['replayed'] >= n0 + 20
    print(f'plain G.synthesis loop, full size, graph-replayed: {ms:.2f} ms/step')
    from inv3d_amd import _lib as _L
    assert ms <= (20.0 if _L.DETERMINISTIC else 8.0), f'{ms:.2f} ms per step'          # (the deterministic build's step is ~2.2x: DESIGN.md 3.5)
    with torch.no_grad():
        for b, s in zip(bufs, saved):
            b.copy_(s)
    ms_e, w_b, l_b = run(False, 8, 2, pinned)
    print(f'per-launch path: {ms_e:.2f} ms/step')
    with torch.no_grad():
        for b, s in zip(bufs, saved):
            b.copy_(s)
    _, w_c, l_c = run(True, 8, 2, pinned)  # same steps and sampling uniforms as the per-launch run, for the trajectory comparison
    assert float((w_c - w_b).abs().max()) < 1e-4 and abs(l_c - l_b) <= 1e-4 * abs(l_b)
    for b in bufs:
        b.requires_grad = False
