"""Forward + image-gradient time of the perceptual-loss networks at the sizes the inversion loops use them."""
import sys
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import loss_nets as LN

dev = 'cuda'


def bench(name, net, img, reps=20):
    tf = net(torch.rand_like(img)).detach()
    x = img.clone().requires_grad_(True)

    def step():
        x.grad = None
        (net(x) - tf).square().sum().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / reps:.3f} ms fwd+bwd', flush=True)


bench('VGG16-LPIPS 256^2', LN.VGG16LPIPS().to(dev), torch.rand(1, 3, 256, 256, device=dev) * 255)
bench('VGG16 features[:15] 512^2', LN.VGG16Features().to(dev), torch.rand(1, 3, 512, 512, device=dev) * 2 - 1)
bench('LPIPS-Alex 512^2', LN.LPIPSAlex().to(dev), torch.rand(1, 3, 512, 512, device=dev) * 2 - 1)
bench('LPIPS-Alex 128^2', LN.LPIPSAlex().to(dev), torch.rand(1, 3, 128, 128, device=dev) * 2 - 1)

# in-loop pose estimator: forward + backward into all 21.9 M parameters + Adam, on the 512^2 target (w_projector.py:122,148,249-261)
from inv3d_amd.pose_net import resnet34_pose
net = resnet34_pose(4).to(dev).requires_grad_(True)
opt = torch.optim.Adam(net.parameters(), lr=1e-6)
img = torch.rand(1, 3, 512, 512, device=dev) * 2 - 1


def pstep():
    opt.zero_grad(set_to_none=True)
    net(img).square().sum().backward()
    opt.step()


for _ in range(3):
    pstep()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    pstep()
e1.record()
torch.cuda.synchronize()
print(f'ResNet-34 pose net 512^2: {e0.elapsed_time(e1) / 20:.3f} ms fwd+bwd+Adam', flush=True)
