"""Which host-side operators launch the small kernels of a Phase-A (latent projection, config C2) step: torch.profiler event tree of one
eager step, written to gpurun_out/phase_a_ops.txt (the graph replays the same launches)."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
proj = LatentProjector(G, target, num_steps=400, cam=cam, seed=100, use_graph=False)
for _ in range(3): proj.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    proj.step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
evs.sort(key=lambda e: e.time_range.start)
def nk(e):
    return len(e.kernels) + sum(nk(c) for c in e.cpu_children)
tot = 0
with open('/root/repo/gpurun_out/phase_a_ops.txt', 'w') as f:
    def dump(e, depth):
        n = nk(e)
        if n == 0: return
        f.write(f"{'  ' * depth}{e.name[:100]}  kernels={n} own={[k.name[:40] for k in e.kernels]}\n")
        if depth < 3:
            for c in e.cpu_children: dump(c, depth + 1)
    for e in evs:
        if e.cpu_parent is None:
            dump(e, 0); tot += nk(e)
print('kernels', tot)
