"""Config C3 with the real-architecture nets (ResNet-34 pose estimator fine-tuned in the loop, VGG16-LPIPS, VGG16 features[:15]) replayed from
its HIP graph, for rocprofv3 --kernel-trace (tools/step_trace.py reads the CSV)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '3dgan-inversion_amd'))
import torch
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
from inv3d_amd.loss_nets import VGG16LPIPS, VGG16Features
from inv3d_amd.pose_net import resnet34_pose
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
P = LatentProjector(G, target, num_steps=400, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=2, seed=1, use_graph=True,
                    pose_net=resnet34_pose(4).to(dev), feature_net=VGG16LPIPS().to(dev), warp_feature_net=VGG16Features().to(dev))
for _ in range(8):
    P.step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    P.step()
torch.cuda.synchronize()
print(f'C3 (real nets) step: {(time.perf_counter() - t) / 20 * 1e3:.2f} ms', 'graph' if P._graph is not None else 'eager')
