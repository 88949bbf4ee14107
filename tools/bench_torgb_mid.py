"""toRGB of the 128^2 / 256^2 backbone blocks: the streaming kernels of csrc/torgb_small.hip (torgb_mid_kernel / torgb_mid_bwd_kernel) against the
implicit GEMM they replace, back-to-back launches (operands warm in the MALL), variants of the epilogue inputs.   python tools/bench_torgb_mid.py"""
import sys, torch, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev = 'cuda'
CL = torch.channels_last
def timeit(f, iters=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (ci, h) in ((256, 128), (128, 256)):
    co = 96
    x = torch.randn(1, ci, h, h, device=dev).contiguous(memory_format=CL)
    w = torch.randn(co, ci, 1, 1, device=dev) / math.sqrt(ci)
    wf, wa = H.pack_weight_fwd(w), H.pack_weight_adj(w)
    s = torch.rand(1, ci, device=dev) + 0.5; b = torch.randn(co, device=dev)
    skip = torch.randn(1, co, h, h, device=dev).contiguous(memory_format=CL)
    half = torch.randn(1, co, h // 2, h // 2, device=dev).contiguous(memory_format=CL)
    out = H.zeros_cl(1, co, h, h, dev)
    cls = H.classes_corr(h, h, 1, 1, 0)
    taps = (0.25, 0.75, 0.75, 0.25)
    r = {}
    r['igemm up2'] = timeit(lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, epi=L.EPI_FWD, bias=b, act='linear', gain=1.0, clamp=-1.0, addend=half, addend_up2_taps=taps))
    r['mid up2'] = timeit(lambda: H.torgb_small(x, wf, s, out, bias=b, addend=half, addend_up2_taps=taps))
    r['mid full'] = timeit(lambda: H.torgb_small(x, wf, s, out, bias=b, addend=skip))
    r['mid none'] = timeit(lambda: H.torgb_small(x, wf, s, out, bias=b))
    dy = torch.randn(1, co, h, h, device=dev).contiguous(memory_format=CL)
    add = torch.randn(1, ci, h, h, device=dev).contiguous(memory_format=CL)
    dx = H.empty_cl(1, ci, h, h, dev); ds = torch.zeros(1, ci, device=dev)
    r['igemm bwd'] = timeit(lambda: H.conv_igemm(dy, wa, co, ci, dx, H.classes_corr_adjoint(h, h, 1, 1, 0), epi=L.EPI_BWD, out_scale=s, xin=x, ds=ds, addend=add))
    r['mid bwd'] = timeit(lambda: H.torgb_small_bwd(dy, wa, s, x, dx, ds=ds, addend=add))
    r['mid bwd no addend'] = timeit(lambda: H.torgb_small_bwd(dy, wa, s, x, dx, ds=ds))
    print(f'{ci}->{co} @{h}^2: ' + '  '.join(f'{k} {v:.1f}' for k, v in r.items()), flush=True)
