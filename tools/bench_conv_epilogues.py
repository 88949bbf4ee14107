import sys, torch, time, math
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev='cuda'
def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
def run(n, ci, co, h, k=3):
    torch.manual_seed(0)
    x = torch.randn(n, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, k, k, device=dev) / math.sqrt(ci*k*k)
    wf = H.pack_weight_fwd(w); wa = H.pack_weight_adj(w)
    s = torch.rand(n, ci, device=dev) + 0.5
    d = torch.rand(n, co, device=dev) + 0.5
    b = torch.randn(co, device=dev)
    nz = torch.randn(h, h, device=dev); ns = torch.tensor(0.1, device=dev)
    cls = H.classes_corr(h, h, k, k, k//2)
    flops = 2.0*n*h*h*k*k*ci*co
    out = H.empty_cl(n, co, h, h, dev)
    t0 = timeit(lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s))
    t1 = timeit(lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, epi=L.EPI_FWD, out_scale=d, bias=b, noise=nz, noise_nstride=0, noise_strength=ns, act='lrelu', alpha=0.2, gain=1.414, clamp=256.0))
    g = torch.randn(n, co, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dx = H.empty_cl(n, ci, h, h, dev); ds = torch.zeros(n, ci, device=dev)
    cls_adj = H.classes_corr_adjoint(h, h, k, k, k//2)
    t2 = timeit(lambda: H.conv_igemm(g, wa, co, ci, dx, cls_adj))
    t3 = timeit(lambda: H.conv_igemm(g, wa, co, ci, dx, cls_adj, epi=L.EPI_BWD, out_scale=s, xin=x, ds=ds))
    dsr = torch.zeros(32, n, ci, device=dev)
    t5 = timeit(lambda: H.conv_igemm(g, wa, co, ci, dx, cls_adj, epi=L.EPI_BWD, out_scale=s, xin=x, ds=dsr))
    H.conv_igemm(g, wa, co, ci, dx, cls_adj, epi=L.EPI_BWD, out_scale=s, xin=x, ds=ds.zero_()); dsr.zero_(); H.conv_igemm(g, wa, co, ci, dx, cls_adj, epi=L.EPI_BWD, out_scale=s, xin=x, ds=dsr)
    print(f'rep32 {t5:6.3f} err {float((dsr.sum(0)-ds).abs().max()/ds.abs().max()):.1e}', end=' ')
    t4 = timeit(lambda: H.conv_igemm(g, wa, co, ci, dx, cls_adj, epi=L.EPI_BWD, out_scale=s))
    print(f'noDS {t4:6.3f}', end=' ')
    print(f'{ci:4d}->{co:4d} @{h:4d}^2: store {t0:6.3f} ms {flops/t0/1e9:6.1f} TF | EPI_FWD {t1:6.3f} {flops/t1/1e9:6.1f} | adj store {t2:6.3f} {flops/t2/1e9:6.1f} | EPI_BWD {t3:6.3f} {flops/t3/1e9:6.1f}')
run(1,128,128,512); run(1,256,256,256); run(1,64,64,512); run(1,512,512,128)
