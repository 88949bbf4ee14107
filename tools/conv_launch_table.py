"""Every implicit-GEMM conv launch of one eager C2 step, in order: geometry, epilogue, time, algorithmic TFLOP/s (in-situ, HIP events)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S, hipops as H
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
proj = LatentProjector(G, target, num_steps=400, cam=cam, seed=100); proj.preheat = 0
for _ in range(3): proj.step()
reps = 5
profs = []
for _ in range(reps):
    prof = H.LaunchProfiler(keep_meta=True); H.PROFILER = prof
    proj.step(); torch.cuda.synchronize(); H.PROFILER = None
    profs.append(prof)
EPI = {0: 'store', 1: 'atomic', 2: 'fwd', 3: 'bwd'}
PREC = {0: 'f32', 1: 'bf16x6', 2: 'bf16x3', 3: 'f16x3'}
tot = 0.0
print(f'{"#":>3} cfg {"in":>14} {"out":>14} taps           epi    ks prec   {"GF":>7} {"us":>7} {"TF/s":>6}')
for i, (rec, m) in enumerate(zip(profs[0].records, profs[0].meta)):
    us = sorted(p.records[i][2].elapsed_time(p.records[i][3]) for p in profs)[reps // 2] * 1e3
    tot += us
    print(f'{i:3d} {rec[0][0]:3d} {m["Hi"]:4d}x{m["Wi"]:<4d}x{m["Ck"]:<4d} {m["Ho"]:4d}x{m["Wo"]:<4d}x{m["Nc"]:<4d} {str(m["taps"]):14s} {EPI[m["epi"]]:6s} {m["ksplit"]:2d} {PREC[m["prec"]]:6s} '
          f'{rec[1] / 1e9:7.2f} {us:7.1f} {rec[1] / us / 1e6:6.1f}')
print(f'total {tot / 1e3:.3f} ms over {len(profs[0].records)} launches')
