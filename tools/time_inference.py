"""Inference consumers at full size: orbit frames/s, 512^3 density grid, mean-latent statistics."""
import sys
import time
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import inference as INF, synthetic as S

dev = torch.device('cuda')
G = S.make_generator(device=dev)
S.load_synthetic_weights(G, seed=0)
ws = S.synth_ws(14, 512, 1, seed=3).to(dev)


def timed(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


list(INF.render_orbit(G, ws, num_frames=4))
dt, frames = timed(lambda: [f for f in INF.render_orbit(G, ws, num_frames=60)])
print(f'orbit: {60 / dt:.1f} frames/s (512^2, backbone cached; {dt / 60 * 1e3:.2f} ms/frame)', flush=True)
INF.density_grid(G, ws, res=64)
for res in (256, 512):
    dt, g = timed(lambda: INF.density_grid(G, ws, res=res))
    print(f'density grid {res}^3: {dt * 1e3:.1f} ms = {res ** 3 / dt / 1e9:.2f} G points/s', flush=True)
dt, (w_avg, w_std) = timed(lambda: INF.estimate_w_stats(G, num_samples=10000))
print(f'w_avg over 10000 mapped latents: {dt * 1e3:.1f} ms (w_std {w_std:.4f})', flush=True)
