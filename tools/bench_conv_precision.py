import sys, torch, time, math
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev='cuda'
def run(n, ci, co, h, k=3, iters=20, convT=False, check=False):
    torch.manual_seed(0)
    x = torch.randn(n, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, k, k, device=dev) / math.sqrt(ci*k*k)
    wf = H.pack_weight_fwd(w)
    s = torch.rand(n, ci, device=dev) + 0.5
    if convT:
        cls, ho, wo = H.classes_convT(h, h, k, k, 2); kw = dict(out_stride=2)
    else:
        cls = H.classes_corr(h, h, k, k, k//2); ho = wo = h; kw = {}
    flops = 2.0*n*h*h*k*k*ci*co
    ref = None
    if check and not convT:
        ref = torch.nn.functional.conv2d((x*s[:,:,None,None]).double(), w.double(), padding=k//2)
    line = f'N={n} {ci:4d}->{co:4d} @{h:4d}^2 k={k} convT={convT}:'
    for prec in ('f32', 'bf16x6', 'f16x3', 'bf16x3'):
        out = H.empty_cl(n, co, ho, wo, dev)
        f = lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, precision=prec, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/iters
        err = ''
        if ref is not None:
            err = f' err={(out.double()-ref).abs().max().item()/ref.abs().max().item():.2e}'
        line += f'  {prec}: {ms:7.3f} ms {flops/ms/1e9:6.1f} TF{err}'
    print(line)
run(1,128,128,128,check=True); run(1,512,512,32,check=True)
run(1,128,128,512); run(1,256,256,256); run(1,256,128,256,convT=True); run(1,512,512,64); run(1,512,512,32); run(1,32,256,128,convT=True); run(1,128,96,256,k=1); run(1,512,256,64,convT=True)
