"""Config C3 step at full size: C2 + pose chain + second (canonical, no-grad) forward + depth-reprojection warping loss.
Variants: free quaternion + stub features | ResNet-34 pose estimator + VGG16-LPIPS + VGG16 features[:15] (random weights)."""
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


def run(name, **kw):
    P = LatentProjector(G, target, num_steps=400, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=2, seed=1, **kw)
    for _ in range(6):
        P.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10
    for _ in range(n):
        P.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f'C3 step, {name}: {dt * 1e3:.2f} ms ({1 / dt:.1f} steps/s)', flush=True)


run('free quaternion, stub feature nets')
run('free quaternion, stub feature nets, HIP graph', use_graph=True)
from inv3d_amd.loss_nets import VGG16LPIPS, VGG16Features
from inv3d_amd.pose_net import resnet34_pose
run('ResNet-34 pose net + VGG16-LPIPS + VGG16 features[:15]', pose_net=resnet34_pose(4).to(dev), feature_net=VGG16LPIPS().to(dev),
    warp_feature_net=VGG16Features().to(dev))
run('ResNet-34 pose net + VGG16-LPIPS + VGG16 features[:15], HIP graph', pose_net=resnet34_pose(4).to(dev), feature_net=VGG16LPIPS().to(dev),
    warp_feature_net=VGG16Features().to(dev), use_graph=True)
