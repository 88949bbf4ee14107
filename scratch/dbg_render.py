import sys, torch, math
sys.path.insert(0,'.'); sys.path.insert(0,'3dgan-inversion_amd')
from oracle import eg3d_oracle as O
from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
from inv3d_amd.training.triplane import OSGDecoder
DEV='cuda'
cfg = O.full_config(); opts = dict(cfg.rendering); res, n = 10, 2
P = O.synth_params(O.small_config(), seed=7)
g = torch.Generator().manual_seed(21)
planes = (torch.randn(n, 3, 32, 64, 64, generator=g) * 0.8)
cam = O.synth_cameras(n, seed=11)
c2w, K = cam[:, :16].reshape(n, 4, 4), cam[:, 16:].reshape(n, 3, 3)
o, dr = O.ray_sampler(c2w, K, res)
u1 = torch.rand(n, res * res, 48, 1, generator=g); u2 = torch.rand(n * res * res, 48, generator=g)
g_rgb = torch.randn(n, res * res, 32, generator=g); g_dep = torch.randn(n, res * res, 1, generator=g)
for which in ('both','rgb','dep'):
    gr_, gd_ = (g_rgb if which!='dep' else g_rgb*0), (g_dep if which!='rgb' else g_dep*0)
    pr, orr, drr = planes.clone().requires_grad_(True), o.clone().requires_grad_(True), dr.clone().requires_grad_(True)
    rgb_r, dep_r, ws_r = O.render(P, pr, orr, drr, opts, u1, u2)
    gr = torch.autograd.grad([rgb_r, dep_r], [pr, orr, drr], [gr_, gd_])
    dec = OSGDecoder(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32}).to(DEV)
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in P.items() if k.startswith('decoder.')})
    R = ImportanceRenderer(); R.set_uniforms(u1.to(DEV), u2.to(DEV))
    pg = planes.to(DEV).requires_grad_(True); og, dg = o.to(DEV).requires_grad_(True), dr.to(DEV).requires_grad_(True)
    rgb, dep, ws = R(pg, dec, og, dg, opts)
    gg = torch.autograd.grad([rgb, dep], [pg, og, dg], [gr_.to(DEV), gd_.to(DEV)])
    e = (gg[1].cpu()-gr[1]).abs().amax(-1).flatten()
    top = torch.topk(e, 5)
    print(which, 'd_origins max err', float(e.max()), 'scale', float(gr[1].abs().max()), 'n rays err>1e-3:', int((e>1e-3).sum()), 'of', e.numel(), 'top', top.values.tolist(), top.indices.tolist())
    e2 = (gg[0].cpu()-gr[0]).abs()
    print('   d_planes max err', float(e2.max()), 'scale', float(gr[0].abs().max()))
