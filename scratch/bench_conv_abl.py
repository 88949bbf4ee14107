import sys, torch, math
sys.path.insert(0,'.'); sys.path.insert(0,'3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev='cuda'
def run(n, ci, co, h, gain):
    k=3
    x = torch.randn(n, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, k, k, device=dev) / math.sqrt(ci*k*k)
    wf = H.pack_weight_fwd(w); s = torch.rand(n, ci, device=dev) + 0.5
    cls = H.classes_corr(h, h, k, k, 1); out = H.empty_cl(n, co, h, h, dev)
    f = lambda: H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, gain=gain)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print(f'abl={gain} {ci}->{co}@{h}: {ms:.3f} ms {2.0*n*h*h*9*ci*co/ms/1e9:.1f} TF')
for g in (1.0, 101.0, 102.0, 103.0):
    run(1,128,128,512,g); run(1,256,256,256,g)
