"""Host-side cost of one Phase-B step (the eager step is host-bound on slow hosts): wall time of forward / objective / backward / optimizer
with a device sync after each, then cProfile's top entries over 10 steps."""
import sys, time, cProfile, pstats, io, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch.nn.functional as F
from inv3d_amd import synthetic as S, hipops as H
from inv3d_amd.inversion import PivotalTuner
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
t = PivotalTuner(G, target, S.synth_ws(14, 512, 1, seed=5).to(dev), cam)
for _ in range(3): t.step()
sync = torch.cuda.synchronize
acc = [0.0] * 4
for _ in range(10):
    if t._arena is None: t._arena = H.ZeroArena(dev)
    with H.zero_arena(t._arena):
        sync(); a = time.perf_counter()
        out = G.synthesis(t.w_pivot[:, :G.backbone.num_ws], t.cam[:, :25], **t.synth_kwargs)
        sync(); b = time.perf_counter()
        loss, _ = t._fused_objective(out)
        sync(); c = time.perf_counter()
        t.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        sync(); d = time.perf_counter()
        t.optimizer.step()
        sync(); e = time.perf_counter()
    for i, v in enumerate((b - a, c - b, d - c, e - d)): acc[i] += v / 10
print('forward %.2f ms, objective %.2f ms, backward %.2f ms, optimizer %.2f ms' % tuple(v * 1e3 for v in acc))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): t.step()
sync(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
