import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch, torch.nn.functional as F
from inv3d_amd import synthetic as S, hipops as H
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
mode = sys.argv[1]
ws1 = S.synth_ws(14, 512, 1, seed=7).to(dev)
G.requires_grad_(False)
for key, ov in (('f16x3', None), ('f16x1', 'f16x1' if 'ov' in mode else None)):
    w = ws1.clone().requires_grad_(True)
    with H.modconv_override(ov):
        o = G.synthesis(w, cam[:1], noise_mode='const', force_fp32=True)
        l = (F.avg_pool2d(o['image'], 2) - F.avg_pool2d(target[:1], 2)).square().sum()
        if 'bw' in mode: l.backward()
        else: torch.autograd.grad(l, [w])
print('eager passes done', flush=True)
for flag in (False, True):
    pr = LatentProjector(G, target, num_steps=400, cam=cam, seed=321, use_graph=True, modconv_f16x1=flag)
    pr.preheat = 0
    for i in range(5):
        o = pr.step()
    torch.cuda.synchronize(); print('projector', flag, float(o['loss']), 'graph' if pr._graph is not None else 'eager', flush=True)
print('OK')
