"""GPU parity of the in-loop pose estimator: ResNetPose on the HIP conv kernels (forward, data gradient,
weight gradient, folded eval-mode BatchNorm) vs oracle/pose_net_oracle.py, which is pinned against the reference's own ResNet class."""
import pytest
import torch

from oracle import pose_net_oracle as PO

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


@pytest.mark.parametrize('dims', [4, 6])
def test_pose_net_matches_oracle_and_golden(dims, golden):
    from inv3d_amd.pose_net import resnet34_pose
    d = golden('pose_net')
    sd = PO.synth_state(seed=3, output_dims=dims)
    net = resnet34_pose(dims)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).requires_grad_(True)
    img = torch.from_numpy(d[f'd{dims}_img'])
    gy = torch.from_numpy(d[f'd{dims}_gy'])
    y = net(img.to(DEV))
    assert _rel(y, torch.from_numpy(d[f'd{dims}_y'])) < 1e-4          # the reference's own output
    y.backward(gy.to(DEV))
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    yo = PO.forward(sdo, img)
    yo.backward(gy)
    worst = {}
    for k, p in net.named_parameters():
        worst[k] = _rel(p.grad, sdo[k].grad)
    # ReLU / max-pool routing can flip for activations that agree to ~1e-7 (see test_gpu_lossnets.close_most): nearly all parameter
    # gradients must match tightly, none may be far off
    bad = [k for k, e in worst.items() if e > 2e-3]
    assert len(bad) <= 0.05 * len(worst), (bad[:5], [worst[k] for k in bad[:5]])
    assert max(worst.values()) < 0.2, max(worst.items(), key=lambda kv: kv[1])


def test_pose_net_refuses_training_mode_batchnorm():
    from inv3d_amd.pose_net import resnet34_pose
    net = resnet34_pose(4)
    with pytest.raises(NotImplementedError):
        net.train()


def test_projector_fine_tunes_the_pose_net():
    """Config C3 with the in-loop estimator: the rotation comes from ResNetPose(target) every step, Adam updates all its parameters
    (w_projector.py:122,148-150,249-261); the step matches the free-quaternion path when the network outputs that quaternion."""
    from inv3d_amd import synthetic as S
    from inv3d_amd.inversion import LatentProjector
    from inv3d_amd.pose_net import resnet34_pose
    from oracle import eg3d_oracle as O
    cfg = O.small_config()
    G = S.make_generator(w_dim=32, z_dim=32, plane_res=32, channel_base=256, channel_max=16, nrr=16, sr_in_res=16, sr_widths=(16, 8),
                         rendering_kwargs=cfg.rendering, device=DEV)
    S.load_synthetic_weights(G, 0)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    target = torch.tanh(torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(7))).to(DEV)
    net = resnet34_pose(4)
    net.load_state_dict(PO.synth_state(seed=3, output_dims=4))
    net = net.to(DEV)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    kw = dict(num_steps=10, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=1, cam_lr=1e-4, seed=3,
              synth_kwargs=dict(render_uniforms=(u1.to(DEV), u2.to(DEV))))
    P = LatentProjector(G, target, pose_net=net, **kw)
    with torch.no_grad():
        q0 = net(target).clone()
    outs = [P.step() for _ in range(3)]
    assert all(torch.isfinite(o['loss']) for o in outs)
    moved = sum(float((p.detach() - before[k]).abs().max()) > 0 for k, p in net.named_parameters())
    assert moved > 100, moved                                   # conv, BatchNorm affine and head parameters all receive gradients
    # same first step as the free quaternion initialised at the network's prediction
    Q = LatentProjector(G, target, **kw)
    with torch.no_grad():
        Q.quat.copy_(q0)
    o = Q.step()
    assert abs(float(o['loss']) - float(outs[0]['loss'])) <= 1e-4 * abs(float(o['loss']))


def test_e4e_encoder_vs_reference(golden):
    """The one-shot latent initialiser (row f2): inv3d_amd.e4e.Encoder4Editing loads the reference's state dict (same keys) and reproduces
    the reference class's 18 codes on the fixture input; PSPEncoder returns code 0 -- what w_projector.py:100 adds to w_avg."""
    import numpy as np
    from inv3d_amd.e4e import Encoder4Editing, PSPEncoder
    from oracle import e4e_oracle as EO
    d = golden('e4e')
    sd = EO.synth_state(seed=5)
    net = Encoder4Editing(50, 'ir_se')
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    img = torch.from_numpy(np.asarray(d['img'])).to(DEV)
    ref = torch.from_numpy(np.asarray(d['codes']))
    y = net(img).cpu()
    scale = float(ref.abs().max())
    assert float((y - ref).abs().max()) <= 2e-4 * scale, float((y - ref).abs().max()) / scale
    psp = PSPEncoder()
    psp.encoder.load_state_dict(sd)
    w0 = psp.to(DEV)(img).cpu()
    assert w0.shape == (2, 512) and float((w0 - ref[:, 0]).abs().max()) <= 2e-4 * scale
