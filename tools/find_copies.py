"""Which Python call sites of one eager C3 step issue device-to-device copies / memsets (they become memcpy / memset NODES in a captured graph and split it)."""
import sys, traceback, collections
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
P = LatentProjector(G, target, num_steps=400, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=2, seed=1, use_graph=False)
G.graph_eager = False
for _ in range(4):
    P.step()
torch.cuda.synchronize()
sites = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('copy_', 'clone', 'zero_', 'zeros', '_to_copy', 'contiguous', 'fill_')):
            fr = [f for f in traceback.extract_stack() if '/inv3d_amd/' in f.filename]
            t = args[0] if args and torch.is_tensor(args[0]) else None
            shape = tuple(t.shape) if t is not None else None
            sites[(name, shape, f'{fr[-1].filename.split("/")[-1]}:{fr[-1].lineno}' if fr else '?')] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    P.step()
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(v, k)
