"""ATen ops of one eager C3 step (which ops are the ~350 tiny launches / the memcpy nodes): torch.profiler key averages."""
import sys
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from torch.profiler import profile, ProfilerActivity
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
P = LatentProjector(G, target, num_steps=400, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=2, seed=1, use_graph=False)
G.graph_eager = False
for _ in range(5):
    P.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    P.step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') or 'Memcpy' in e.key or 'memcpy' in e.key.lower()]
rows.sort(key=lambda e: -e.count)
for e in rows[:70]:
    print(f'{e.count:4d}  {e.key:40s} {str(e.input_shapes)[:110]}')
