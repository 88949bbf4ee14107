"""1x1 toRGB-shaped launches of the implicit-GEMM kernel (memory-bound): forward with skip addend, data gradient with style reduction."""
import sys, torch, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev = 'cuda'
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (ci, co, h) in ((128, 96, 256), (256, 96, 128), (512, 96, 64), (128, 3, 512), (256, 3, 256)):
    cop = (co + 3) // 4 * 4
    x = torch.randn(1, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, device=dev) / math.sqrt(ci)
    wf, wa = H.pack_weight_fwd(w), H.pack_weight_adj(w)
    if cop != co:
        wp = torch.zeros(ci, cop, device=dev); wp[:, :co] = wa; wa = wp
    s = torch.rand(1, ci, device=dev) + 0.5; b = torch.randn(co, device=dev)
    skip = torch.randn(1, cop, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    out = H.zeros_cl(1, cop, h, h, dev)
    cls = H.classes_corr(h, h, 1, 1, 0)
    tf = timeit(lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, epi=L.EPI_FWD, bias=b, act='linear', gain=1.0, clamp=-1.0, addend=skip))
    dy = torch.randn(1, cop, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dx = H.empty_cl(1, ci, h, h, dev); ds = torch.zeros(1, ci, device=dev)
    tb = timeit(lambda: H.conv_igemm(dy, wa, cop, ci, dx, H.classes_corr_adjoint(h, h, 1, 1, 0), epi=L.EPI_BWD, out_scale=s, xin=x, ds=ds))
    bf = (ci + 2 * cop) * h * h * 4 / 1e6; bb = (cop + 2 * ci) * h * h * 4 / 1e6
    print(f'{ci}->{co} @{h}^2: fwd {tf*1e3:6.1f} us ({bf:6.1f} MB, {bf/tf/1e3:5.2f} TB/s)   dgrad {tb*1e3:6.1f} us ({bb:6.1f} MB, {bb/tb/1e3:5.2f} TB/s)')
