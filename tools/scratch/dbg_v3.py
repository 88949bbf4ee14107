import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import hipops as H, _lib as L
DEV='cuda'
g = torch.Generator().manual_seed(41)
for (n, ci, h, w, co) in ((1, 32, 16, 64, 128), (1, 16, 4, 32, 64)):
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wimg = H.split_weight(H.pack_weight_fwd(wt.to(DEV)), co, ci, 9)
    aimg = H.split_activation(xc, H.absmax(xc))
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1).float()
    for plan in ((4, 4), (2, 4), (2, 8)):
        out = H.empty_cl(n, co, h, w, DEV)
        H.conv_v3(aimg, wimg, out, H.classes_corr(h, w, 3, 3, 1), plan=plan, epi=L.EPI_STORE)
        o = out.cpu()
        bad = ~torch.isfinite(o)
        err = (torch.nan_to_num(o) - ref).abs()
        print(plan, (n, ci, h, w, co), 'nonfinite', int(bad.sum()), 'of', o.numel(), 'max err (finite)', float(err[~bad].max()))
        if bad.any():
            idx = bad.nonzero()
            print('  rows', sorted(set(idx[:, 2].tolist()))[:20], 'cols', sorted(set(idx[:, 3].tolist()))[:40], 'nch', len(set(idx[:, 1].tolist())))
        big = (err > 1e-3) & ~bad
        if big.any():
            idx = big.nonzero()
            print('  WRONG rows', sorted(set(idx[:, 2].tolist()))[:20], 'cols', sorted(set(idx[:, 3].tolist()))[:40], 'nch', len(set(idx[:, 1].tolist())))
