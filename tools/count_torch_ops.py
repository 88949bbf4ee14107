"""Count ATen ops launched during one eager C2 step (or, with the argument `b`, one pivotal-tuning step), grouped by op name and
calling source line (where the tiny launches come from)."""
import sys, collections, traceback, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from torch.utils._python_dispatch import TorchDispatchMode
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector, PivotalTuner
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
if len(sys.argv) > 1 and sys.argv[1] == 'b':
    proj = PivotalTuner(G, target, S.synth_ws(14, 512, 1, seed=5).to(dev), cam)
else:
    proj = LatentProjector(G, target, num_steps=400, cam=cam, seed=100); proj.preheat = 0
for _ in range(2): proj.step()
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('view', 'reshape', 'as_strided', 'detach', 'alias', 'expand', 'permute', 'transpose', 'unbind', 'select', 'slice', 'narrow', 'squeeze', 't.default', 'empty', 'is_', 'size', 'stride', 'numel', 'sym_')):
            return func(*args, **(kwargs or {}))
        st = traceback.extract_stack()
        src = next((f'{f.filename.split("/")[-1]}:{f.lineno}' for f in reversed(st) if 'inv3d_amd' in f.filename), 'autograd/other')
        cnt[(name, src)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    proj.step()
torch.cuda.synchronize()
for (name, src), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f'{n:4d} {name:40s} {src}')
print('total', sum(cnt.values()))
