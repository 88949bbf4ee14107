"""The two 4x4 FIR passes of an up-sampling layer: forward epilogue (FIR + demod + noise + bias + lrelu) and the adjoint FIR of the backward."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H
from inv3d_amd.fused import fir44
dev = 'cuda'
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (c, h) in ((128, 512), (256, 256), (256, 128), (512, 64)):
    z = torch.randn(1, c, h + 1, h + 1, device=dev).contiguous(memory_format=torch.channels_last)
    out = H.empty_cl(1, c, h, h, dev)
    d = torch.rand(1, c, device=dev) + 0.5; nz = torch.randn(h, h, device=dev); ns = torch.tensor(0.1, device=dev); b = torch.randn(c, device=dev)
    f = fir44(dev)
    tf = timeit(lambda: H.epilogue_fwd(z, out, fir=f, pad0=1, fir_gain=4.0, d=d, noise=nz, noise_nstride=0, noise_strength=ns, bias=b, act='lrelu', alpha=0.2, gain=1.414, clamp=256.0))
    dz = torch.randn(1, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    tb = timeit(lambda: H.upfirdn2d_nhwc(dz, f, pad=(2, 2, 2, 2), flip=True, gain=4.0))
    mb = 2 * c * h * h * 4 / 1e6
    print(f'C={c} {h}^2: epilogue_fwd {tf*1e3:6.1f} us ({mb/tf/1e3:4.2f} TB/s)   FIR adjoint {tb*1e3:6.1f} us ({mb/tb/1e3:4.2f} TB/s)')
