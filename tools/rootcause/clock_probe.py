#!/usr/bin/env python3
"""Which clock do the matrix kernels run at?  20 launches each of (a) the register-only v_mfma_f32_32x32x16_f16 loop on random data (eg3d_probe_mfma_f16),
(b) the same loop on zeros, (c) the dominant convolution (conv_v2 8-row, 512^2 x 128 -> 128, fused forward epilogue), (d) a memory-bound pass (split_activation),
for   rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace   : cycles / duration = the average shader clock of each launch (the matrix pipe is power-limited)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, '3dgan-inversion_amd'))
import torch
from inv3d_amd import hipops as H, _lib as L
dev = 'cuda'
blocks, iters = 1024, 2000
out = torch.empty(blocks * 256, device=dev)
rnd = (torch.rand(4096 * 8, device=dev) * 2 - 1).mul_(1000.0).half()
zer = torch.zeros(4096 * 8, device=dev).half()
g = torch.Generator().manual_seed(1)
ci = co = 128; h = 512
x = torch.randn(1, ci, h, h, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(dev)
s = (1 + 0.5 * torch.randn(1, ci, generator=g)).to(dev)
d = (0.5 + torch.rand(1, co, generator=g)).to(dev)
wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
cls = H.classes_corr(h, h, 3, 3, 1)
o = H.empty_cl(1, co, h, h, dev)
am = H.absmax(x)
for rep in range(20):
    L.check(L.lib().eg3d_probe_mfma_f16(rnd.data_ptr(), out.data_ptr(), blocks, iters, L.stream_ptr()), 'probe')
for rep in range(20):
    L.check(L.lib().eg3d_probe_mfma_f16(zer.data_ptr(), out.data_ptr(), blocks, iters, L.stream_ptr()), 'probe')
for rep in range(20):
    aimg = H.split_activation(x, am, in_scale=s)
    H.conv_v2(aimg, wimg, o, cls, epi=L.EPI_FWD, out_scale=d, act='lrelu', alpha=0.2, gain=1.4, patch_rows=8)
torch.cuda.synchronize()
