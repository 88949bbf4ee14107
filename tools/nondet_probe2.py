"""Run-to-run differences of the C2 step, autograd node by node: the outputs of every backward of the package's autograd functions are
recorded in two runs from the same state and compared bit for bit, in execution order."""
import os, sys
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S, fused, inversion, loss_nets
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)

REC = None
def patch(mod):
    for name in dir(mod):
        cls = getattr(mod, name)
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and 'backward' in cls.__dict__:
            orig = cls.__dict__['backward'].__func__ if isinstance(cls.__dict__['backward'], staticmethod) else cls.backward
            def wrapped(ctx, *g, _orig=orig, _name=f'{mod.__name__.split(".")[-1]}.{name}'):
                out = _orig(ctx, *g)
                if REC is not None:
                    outs = out if isinstance(out, tuple) else (out,)
                    REC.append((_name, [o.detach().clone() if torch.is_tensor(o) else None for o in outs],
                                [o.detach().clone() if torch.is_tensor(o) else None for o in g]))
                return out
            cls.backward = staticmethod(wrapped)
for m in (fused, inversion, loss_nets):
    patch(m)
from inv3d_amd.training import triplane
patch(triplane)

def run(steps):
    global REC
    torch.manual_seed(123)
    P = LatentProjector(G, target, num_steps=400, seed=1, use_graph=False)
    for s in range(steps):
        REC = [] if s == steps - 1 else None
        P.step()
    torch.cuda.synchronize()
    r = REC; REC = None
    return r, P.w_opt.grad.detach().clone()

run(1)
NS = int(os.environ.get("STEPS", "1")); a, ga = run(NS); b, gb = run(NS)
print('nodes', len(a), len(b), 'w_opt.grad equal:', torch.equal(ga, gb))
for i, ((na, oa, ia), (nb, ob, ib)) in enumerate(zip(a, b)):
    bad_in = [k for k, (u, v) in enumerate(zip(ia, ib)) if u is not None and v is not None and u.shape == v.shape and not torch.equal(u, v)]
    bad = [k for k, (u, v) in enumerate(zip(oa, ob)) if u is not None and v is not None and u.shape == v.shape and not torch.equal(u, v)]
    if bad or bad_in:
        print(f'{i:3d} {na}: inputs differing {bad_in}  outputs differing {bad}  shapes {[tuple(oa[k].shape) for k in bad]}')
from inv3d_amd import _lib as L
print('deterministic build:', bool(L.lib().eg3d_det_enabled()), ' misses:', L.det_misses())
