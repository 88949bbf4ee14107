"""Whole-budget inversion of one synthetic image at full size (BASELINE.json metric, second half: "final PSNR"):
400 latent steps (Phase A) + up to 400 pivotal-tuning steps (Phase B), configs/hyperparameters.py defaults.
The target is the render of a different latent by the same random-weight generator, so the optimum is reachable."""
import argparse
import sys
import time
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S
from inv3d_amd.coach import InversionCoach

ap = argparse.ArgumentParser()
ap.add_argument('--loss-net', default='stub', choices=['stub', 'real'])
ap.add_argument('--steps-a', type=int, default=400)
ap.add_argument('--steps-b', type=int, default=400)
args = ap.parse_args()
dev = torch.device('cuda')
G = S.make_generator(device=dev)
S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
fa = fb = None
if args.loss_net == 'real':
    from inv3d_amd.loss_nets import VGG16LPIPS, LPIPSAlex
    fa, fb = VGG16LPIPS().to(dev), LPIPSAlex().to(dev)






coach = InversionCoach(G, first_inv_steps=args.steps_a, max_pti_steps=args.steps_b, lpips_threshold=0.0, use_graph=True, early_stop_interval=50,
                       feature_net=fa, w_avg_samples=0)      # synthetic weights: the latent distribution is centred on 0 by construction
if fb is not None:
    import inv3d_amd.coach as C
    _PT = C.PivotalTuner
    C.PivotalTuner = lambda *a, **k: _PT(*a, **dict(k, feature_net=fb))
torch.cuda.synchronize()
t0 = time.perf_counter()
r = coach.invert('synthetic', target, cam)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'loss nets: {args.loss_net};  {r.steps_a} latent steps + {r.steps_b} tuning steps in {dt:.2f} s wall '
      f'({(r.steps_a + r.steps_b) / dt:.1f} steps/s incl. set-up, graph capture and two evaluation renders)')
print(f'PSNR after the latent phase {r.psnr_pivot:.2f} dB, after pivotal tuning {r.psnr_tuned:.2f} dB (MSE {r.mse_tuned:.3e})')
