"""The fused toRGB data gradient + activation backward + operand split pass (eg3d_torgb_dgrad_act_split) at the SR head's two sizes, with the
reductions switched off one at a time: which of them costs what (us per launch, TB/s over the algorithmic bytes)."""
import sys, math, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H
DEV = 'cuda'
CL = torch.channels_last
g = torch.Generator().manual_seed(7)


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (c, h, with_add) in ((128, 512, False), (256, 256, True)):
    n, w = 1, h
    dy4 = (torch.randn(n, 4, h, w, generator=g) * 1e-3).to(DEV).contiguous(memory_format=CL)
    wa = (torch.randn(c, 4, generator=g) / 2).to(DEV).contiguous()
    s = (1 + 0.5 * torch.randn(n, c, generator=g)).to(DEV)
    x = (torch.randn(n, c, h, w, generator=g) * 1.5).to(DEV).contiguous(memory_format=CL)
    add = (torch.randn(n, c, h, w, generator=g) * 2e-3).to(DEV).contiguous(memory_format=CL) if with_add else None
    d = (0.5 + torch.rand(n, c, generator=g)).to(DEV)
    bias = (torch.randn(c, generator=g) * 0.1).to(DEV)
    noise = torch.randn(h, w, generator=g).to(DEV)
    st = torch.tensor(0.37, device=DEV)
    dy_amax = H.absmax(dy4)
    add_amax = H.absmax(add) if add is not None else None
    acc = dict(dbias=torch.zeros(c, device=DEV), dd=torch.zeros(n, c, device=DEV), dnoise=torch.zeros_like(noise), dstrength=torch.zeros((), device=DEV))
    ds = torch.zeros(n, c, device=DEV)
    gb = (x.numel() * 4 * (2 + (1 if with_add else 0)) + dy4.numel() * 4) / 1e9
    for name, drop in (('all reductions', ()), ('no dnoise/dstrength', ('dnoise', 'dstrength')), ('no noise at all', ('dnoise', 'dstrength', 'noise')),
                       ('no dd', ('dd',)), ('no dbias/dd/ds', ('dbias', 'dd', 'ds')), ('nothing but dz', ('dbias', 'dd', 'ds', 'dnoise', 'dstrength', 'noise'))):
        kw = {k: (None if k in drop else v) for k, v in acc.items()}
        spec = H.ActBwdSpec(d=d, bias=bias, noise=None if 'noise' in drop else noise, noise_nstride=0, noise_strength=None if 'noise' in drop else st, act='lrelu',
                            alpha=0.2, gain=math.sqrt(2), clamp=256.0, dnoise_nstride=0, **kw)
        t = timeit(lambda: H.torgb_dgrad_act_split(dy4, wa, x, s, spec, dy_amax, ds=None if 'ds' in drop else ds, addend=add, addend_amax=add_amax))
        print(f'{h}^2 x {c}{" +addend" if with_add else ""}: {name:22s} {t:7.1f} us  {gb / t * 1e6 / 1e3:5.2f} TB/s', flush=True)
