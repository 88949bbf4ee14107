"""Which layers still run a stand-alone activation-backward pass (eg3d_modconv_epilogue_bwd) in one eager C2 step, and the element-wise passes next to it."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S, hipops as H, graphed
from inv3d_amd.inversion import LatentProjector
graphed.ENABLED = False
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
proj = LatentProjector(G, target, num_steps=400, cam=cam, seed=100); proj.preheat = 0
for _ in range(2): proj.step()
log = []
def wrap(name):
    orig = getattr(H, name)
    def f(*a, **k):
        t = next((x for x in a if torch.is_tensor(x) and x.dim() == 4), None)
        log.append((name, tuple(t.shape) if t is not None else None, {kk: (vv if not torch.is_tensor(vv) else 'T') for kk, vv in k.items() if kk in ('act', 'clamp', 'fir', 'up', 'down')}))
        return orig(*a, **k)
    setattr(H, name, f)
for nm in ('epilogue_bwd', 'epilogue_fwd', 'split_activation', 'upfirdn2d_nhwc', 'dgrad_finish', 'dgrad_finish_act', 'torgb_dgrad_act', 'conv_v2', 'conv_up2', 'conv_igemm'):
    wrap(nm)
proj.step(); torch.cuda.synchronize()
for e in log: print(e)
