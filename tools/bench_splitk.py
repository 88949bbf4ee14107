"""Split-K sweep for the low-resolution 512-channel layers (EPI_ATOMIC into a zeroed buffer): time and tile configuration per ksplit."""
import sys, math, torch
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
import ctypes as C
dev = 'cuda'
for (ci, co, h) in ((512, 512, 64), (512, 512, 32), (512, 512, 16), (256, 256, 128)):
    x = torch.randn(1, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=dev) / math.sqrt(ci * 9)
    wf = H.pack_weight_fwd(w)
    s = torch.rand(1, ci, device=dev) + 0.5
    cls = H.classes_corr(h, h, 3, 3, 1)
    out = H.zeros_cl(1, co, h, h, dev)
    fl = 2.0 * h * h * 9 * ci * co
    line = f'{ci}->{co} @{h}^2:'
    for ks in (1, 2, 4, 8, 16, 32):
        f = lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, epi=L.EPI_ATOMIC, ksplit=ks, precision='f16x3')
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += f'  ks={ks}: {ms*1e3:6.1f} us {fl/ms/1e9:5.0f} TF'
    print(line, flush=True)
