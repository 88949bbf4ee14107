import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import hipops as H, _lib as L
dev = torch.device('cuda')
ci = co = 512
res = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(0)
w = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(dev)
wimg = H.split_weight(H.pack_weight_fwd(w), co, ci, 9)
s = (1 + 0.5 * torch.randn(1, ci, generator=g)).to(dev)
x = torch.randn(1, ci, res, res, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
ax = H.absmax(x); cls = H.classes_corr(res, res, 3, 3, 1); out = H.empty_cl(1, co, res, res, dev)
logw = H.conv_lr_plan(ci, co, cls, 1, force=True)[0]
def t(ks, n=20):
    for _ in range(3): H.conv_lr(x, ax, wimg, out, cls, (logw, ks), in_scale=s, epi=L.EPI_STORE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); best = 1e9
    for _ in range(n):
        e0.record(); H.conv_lr(x, ax, wimg, out, cls, (logw, ks), in_scale=s, epi=L.EPI_STORE); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
import os
print(os.environ.get('EG3D_LIBNAME', 'base'), 'res', res, ' '.join(f'ks={k}: {t(k):6.1f}' for k in (1, 2, 8)), flush=True)
