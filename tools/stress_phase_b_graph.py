"""Soak test of the graph-replayed pivotal-tuning step: 3000 full-size steps, a device-wide synchronise every 50, an eager early-stop check every 97
(gpurun: `timeout 600 python tools/stress_phase_b_graph.py`; round 2: 8.97 ms/step, loss 0.033 -> 0.020, no fault)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import PivotalTuner
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
t = PivotalTuner(G, target, S.synth_ws(14, 512, 1, seed=5).to(dev), cam, lpips_threshold=0.0, use_graph=True)
t0 = time.perf_counter()
for i in range(3000):
    r = t.step(early_stop=(i % 97 == 96))
    if i % 50 == 49: torch.cuda.synchronize()
    if i % 500 == 499: print(i, float(r['loss']), flush=True)
torch.cuda.synchronize()
print('ok', (time.perf_counter() - t0) / 3000 * 1e3, 'ms/step', t._graph is not None)
