"""Weight-gradient implicit GEMM: time vs the number of cell slices (fp32-atomic accumulation of the slices), per arithmetic."""
import sys, torch, math
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
precs = sys.argv[1].split(',') if len(sys.argv) > 1 else ['f32', 'f16x3', 'f16x1']
splits = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 32, 64, 128, 256]
for (ci, co, h) in ((128, 128, 512), (256, 256, 256), (128, 128, 256), (512, 512, 64), (512, 512, 16)):
    x = torch.randn(1, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    g = torch.randn(1, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    s = torch.rand(1, ci, device=dev) + 0.5
    dw = torch.zeros(co, 9 * ci, device=dev)
    cls = H.classes_corr(h, h, 3, 3, 1)
    fl = 2.0 * h * h * 9 * ci * co
    amax = g.abs().max().reshape(1)
    for prec in precs:
        line = f'{ci}->{co} @{h}^2 {prec}:'
        for ps in splits:
            t = timeit(lambda: H.conv_wgrad(x, g, ci, co, dw, cls, in_scale=s, psplit=ps, precision=prec, g_amax=amax if prec != 'f32' else None))
            line += f'  ps={ps}: {t*1e3:6.1f} us {fl/t/1e9:5.0f} TF'
        print(line)
