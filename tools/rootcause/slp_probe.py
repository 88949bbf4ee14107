#!/usr/bin/env python3
"""Where the forward epilogue of conv_v2 goes wrong when the library is built WITH the SLP vectoriser: the fused launch against the same
launch with the plain-store epilogue + the epilogue in torch, per patch height; histogram of the wrong elements over (float4 component, column group, patch row,
patch column) and which term of  lrelu(z d + noise s + b) g  is missing.   EG3D_LIBNAME=libeg3d_hip_slp.so python tools/rootcause/slp_probe.py"""
import collections
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, '3dgan-inversion_amd'))
import torch                                             # noqa: E402
from inv3d_amd import hipops as H, _lib as L             # noqa: E402
DEV = 'cuda'
H.V2_KHALVES = False
for (ci, h, co, rows) in ((128, 512, 128, 8), (128, 256, 128, 4), (256, 128, 256, 4)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, ci, h, h, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(1, ci, generator=g)).to(DEV)
    d = (0.5 + torch.rand(1, co, generator=g)).to(DEV)
    noise, strength = torch.randn(h, h, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    aimg = H.split_activation(x, H.absmax(x), in_scale=s)
    wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
    cls = H.classes_corr(h, h, 3, 3, 1)
    z = H.empty_cl(1, co, h, h, DEV)
    H.conv_v2(aimg, wimg, z, cls, epi=L.EPI_STORE, patch_rows=rows)
    pre = z * d[:, :, None, None] + noise * 0.3 + bias[None, :, None, None]
    ref = torch.nn.functional.leaky_relu(pre, 0.2) * 1.4
    pre_nonoise = z * d[:, :, None, None] + bias[None, :, None, None]
    ref_nonoise = torch.nn.functional.leaky_relu(pre_nonoise, 0.2) * 1.4
    hist = collections.Counter(); nbad = 0; nmiss = 0
    for it in range(20):
        out, am = H.empty_cl(1, co, h, h, DEV), torch.zeros(1, device=DEV)
        H.conv_v2(aimg, wimg, out, cls, epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength, act='lrelu', alpha=0.2, gain=1.4,
                  clamp=-1.0, out_amax=am, patch_rows=rows)
        bad = ((out - ref).abs() > 1e-4).nonzero()
        nbad += len(bad)
        if len(bad):
            b = bad[:, 1:]
            nmiss += int(((out - ref_nonoise).abs()[0, b[:, 0], b[:, 1], b[:, 2]] < 1e-5).sum())
            for c, y, xx in b[:4000].tolist():
                hist[('comp', c % 4)] += 1; hist[('c4>=16', (c % 128) // 4 >= 16)] += 1; hist[('row', y % rows)] += 1; hist[('x&1', xx & 1)] += 1; hist[('tile_n', c // 128)] += 1
                hist[('unit', ((y % rows) % (rows // 2)) )] += 1
    print(f'{os.path.basename(L.LIB_PATH)} {h}^2 x {ci} -> {co} rows {rows}: wrong elements {nbad} in 20 launches; of those equal to the epilogue WITHOUT the noise term: {nmiss}')
    for k in sorted(hist, key=str): print('    ', k, hist[k])
