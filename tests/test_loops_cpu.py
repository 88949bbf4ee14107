"""BASELINE.json configs[0] (the reference's own CPU-runnable case): a few projector steps + 10 PTI steps on CPU through the
pure-PyTorch path -- here the oracle (small generator of the same topology, so the whole thing takes seconds).  Plumbing check:
the loops run, gradients reach every optimised tensor, the loss goes down."""
import torch

from oracle import eg3d_oracle as O
from oracle import inversion_oracle as IO


def _setup():
    cfg = O.small_config()
    P = O.synth_params(cfg, 0)
    cam = O.synth_cameras(1, seed=2)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    with torch.no_grad():
        target = O.synthesis(P, cfg, O.synth_ws(cfg, 1, seed=3), cam, u1, u2, noise_mode='const')['image'].clamp(-1, 1)
    return cfg, P, cam, u1, u2, target


def test_projector_steps_cpu():
    cfg, P, cam, u1, u2, target = _setup()
    init_noise = {k: O._randn('init.' + k, 9, v.shape) for k, v in P.items() if k.endswith('noise_const')}
    proj = IO.ProjectorOracle(P, cfg, target, num_steps=20, cam=cam, init_noise=init_noise, w_start=O.synth_ws(cfg, 1, seed=1)[:, :1])
    w0 = proj.w_opt.detach().clone()
    dists = []
    for i in range(8):
        dists.append(float(proj.step(u1, u2, w_noise=None)['dist']))
    assert all(map(lambda v: v == v, dists))
    assert min(dists[4:]) < dists[0], dists              # the feature distance goes down once the lr ramp has started
    assert float((proj.w_opt.detach() - w0).abs().max()) > 0
    assert proj.w_opt.grad is not None and all(b.grad is not None for b in proj.bufs[:13])


def test_pti_10_steps_cpu():
    cfg, P, cam, u1, u2, target = _setup()
    w_pivot = O.synth_ws(cfg, 1, seed=1)
    tuner = IO.PivotalTunerOracle(P, cfg, target, w_pivot, cam)
    losses = [float(tuner.step(u1, u2, noise_mode='const')['loss']) for _ in range(10)]
    assert losses[-1] < losses[0]
    n_grad = sum(1 for p in tuner.params if p.grad is not None and float(p.grad.abs().max()) > 0)
    assert n_grad >= len(tuner.params) - 12       # mapping network (6 tensors) and unused SR noise strengths get no gradient
    psnr = float(O.psnr_01(tuner.last['image'], target))
    assert psnr == psnr


def test_config_c1_at_full_size_cpu():
    """BASELINE.json configs[0] as worded: the ffhqrebalanced512-128-shaped generator, one image, 10 PTI steps on CPU through the pure-PyTorch
    path (the oracle = the pinned port of the reference's `_ref` ops) -- preceded by 2 latent steps so both phases of PTI run at full size.
    ~1 minute on 8 cores; bench.py times the same steps on the GPU box's host as `cpu_baseline` / `cpu_baseline_c1`."""
    cfg = O.full_config()
    P = O.synth_params(cfg, 0)
    cam = O.synth_cameras(1, seed=2)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    w0 = O.synth_ws(cfg, 1, seed=1)
    proj = IO.ProjectorOracle(P, cfg, target, num_steps=400, cam=cam, w_start=w0[:, :1])
    d = [float(proj.step(u1, u2)['dist']) for _ in range(2)]
    assert all(v == v and v > 0 for v in d)
    assert proj.w_opt.grad is not None and float(proj.w_opt.grad.abs().max()) > 0
    assert all(b.grad is not None for b in proj.bufs + proj.bufs2)         # backbone AND SR maps are optimised leaves (w_projector.py:120-131)
    w_pivot = proj.w_opt.detach().repeat(1, cfg.num_ws, 1)
    tuner = IO.PivotalTunerOracle(P, cfg, target, w_pivot, cam)
    losses = [float(tuner.step(u1, u2, noise_mode='const')['loss']) for _ in range(10)]
    assert all(v == v for v in losses) and losses[-1] < losses[0], losses
    assert tuple(tuner.last['image'].shape) == (1, 3, 512, 512)
    n_grad = sum(1 for p in tuner.params if p.grad is not None and float(p.grad.abs().max()) > 0)
    assert n_grad >= len(tuner.params) - 12
