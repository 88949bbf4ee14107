"""conv_wgrad_v2 (split images, LDS-DMA + transposing LDS reads) vs the loader-split weight-gradient kernel, per layer geometry and arithmetic."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H
dev = 'cuda'
def timeit(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (ci, co, h) in ((128, 128, 512), (256, 256, 256), (128, 128, 256), (256, 256, 128), (512, 512, 64)):
    x = torch.randn(1, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    g = (torch.randn(1, co, h, h, device=dev) * 1e-5).contiguous(memory_format=torch.channels_last)
    s = torch.rand(1, ci, device=dev) + 0.5
    dw = torch.zeros(co, 9 * ci, device=dev)
    cls = H.classes_corr(h, h, 3, 3, 1)
    fl = 2.0 * h * h * 9 * ci * co
    amax = g.abs().max().reshape(1)
    ximg = H.split_activation(x, H.absmax(x), in_scale=s)
    gimg = H.split_activation(g, amax)
    for prec, prod in (('f16x3', 3), ('f16x1', 1)):
        t0 = timeit(lambda: H.conv_wgrad(x, g, ci, co, dw, cls, in_scale=s, psplit=0, precision=prec, g_amax=amax))
        line = f'{ci}->{co} @{h}^2 {prec}: old {t0*1e3:6.1f} us {fl/t0/1e9:4.0f} TF |'
        for rg in (0, 4, 8, 16, 32, 64):
            if rg > h // 2: continue
            t = timeit(lambda: H.conv_wgrad_v2(gimg, ximg, dw, cls, products=prod, row_groups=rg))
            line += f' rg={rg}: {t*1e3:6.1f} us {fl/t/1e9:4.0f} TF'
        print(line, flush=True)
