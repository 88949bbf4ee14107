import sys, torch, time, math
sys.path.insert(0,'.'); sys.path.insert(0,'3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev='cuda'
def run(n, ci, co, h, k=3, iters=30, convT=False, ks=1):
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
    out = H.zeros_cl(n, co, ho, wo, dev)
    f = lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, ksplit=ks, epi=L.EPI_ATOMIC if ks > 1 else L.EPI_STORE, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/iters
    print(f'N={n} {ci:4d}->{co:4d} @{h:4d}^2 convT={convT} ks={ks}: {ms:7.3f} ms {flops/ms/1e9:6.1f} TF')
for ks in (1,2,3,4,6):
    run(1,512,512,64,ks=ks)
for ks in (1,2,4,8):
    run(1,512,512,32,ks=ks)
for ks in (1,2,4):
    run(1,512,256,64,convT=True,ks=ks)
