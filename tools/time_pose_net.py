"""In-loop pose estimator step (forward + backward into all parameters + Adam) at 256^2, for rocprofv3 --stats."""
import sys
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import time, torch
from inv3d_amd.pose_net import resnet34_pose
dev = 'cuda'
net = resnet34_pose(4).to(dev).requires_grad_(True)
opt = torch.optim.Adam(net.parameters(), lr=1e-6, fused=True)
img = torch.rand(1, 3, 256, 256, device=dev) * 255          # what the projector feeds it: the 0..255 target at 256^2 (w_projector.py:106-110,148)
def step():
    opt.zero_grad(set_to_none=True)
    net(img).square().sum().backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(f'pose net step: {(time.perf_counter() - t) * 100:.2f} ms', flush=True)
