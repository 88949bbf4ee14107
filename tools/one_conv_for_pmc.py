import sys, torch, math
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/3dgan-inversion_amd')
from inv3d_amd import hipops as H, _lib as L
dev='cuda'
n,ci,co,h,k=1,128,128,512,3
x = torch.randn(n, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, k, k, device=dev) / math.sqrt(ci*k*k)
wf = H.pack_weight_fwd(w); s = torch.rand(n, ci, device=dev) + 0.5
cls = H.classes_corr(h, h, k, k, k//2); out = H.empty_cl(n, co, h, h, dev)
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
for _ in range(10): H.conv_igemm(x, wf, ci, co, out, cls, in_scale=s, precision=prec)
torch.cuda.synchronize()
