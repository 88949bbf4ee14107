import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
proj = LatentProjector(G, target, num_steps=2000, cam=cam, seed=100, use_graph=True)
t0 = time.time()
for i in range(2000):
    out = proj.step()
    if i % 500 == 499:
        torch.cuda.synchronize()
        print(i, float(out['dist']), float(out['loss']), flush=True)
torch.cuda.synchronize()
print('ok', (time.time() - t0) / 2000 * 1e3, 'ms/step', bool(torch.isfinite(out['loss'])), float(proj.optimizer.step_t))
