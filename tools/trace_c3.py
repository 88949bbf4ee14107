"""Config C3 step (free quaternion, stub nets) replayed from its HIP graph, for rocprofv3 --kernel-trace:  tools/step_trace.py reads the CSV."""
import sys, time
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
P = LatentProjector(G, target, num_steps=400, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=2, seed=1, use_graph=True)
for _ in range(8):
    P.step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    P.step()
torch.cuda.synchronize()
print(f'C3 step: {(time.perf_counter() - t) / 20 * 1e3:.2f} ms', 'graph' if P._graph is not None else 'eager')
