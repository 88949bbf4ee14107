"""Phase-B (pivotal tuning, config C4) step time on the full-size generator: forward + backward into ALL generator weights + Adam."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import synthetic as S, hipops as H
from inv3d_amd.inversion import PivotalTuner
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
w_pivot = S.synth_ws(14, 512, 1, seed=5).to(dev)
tuner = PivotalTuner(G, target, w_pivot, cam, use_graph='graph' in sys.argv[1:])
for _ in range(4): tuner.step()
torch.cuda.synchronize(); t = time.perf_counter()
n = 40 if tuner.use_graph else 10
for _ in range(n): tuner.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f'Phase B step: {dt*1e3:.2f} ms  ({1/dt:.1f} steps/s)', 'graph' if tuner._graph is not None else 'eager', tuner.graph_capture_error)
