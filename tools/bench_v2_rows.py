"""Pre-split 3x3 conv (eg3d_conv2d_v2) at 8 / 4 / 2-row patches against the loader-split kernel on the backbone's under-filled grids
(fused forward epilogue in all of them): time per launch, TFLOP/s, and agreement between the patch heights."""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import hipops as H, _lib as L
DEV = 'cuda'
g = torch.Generator().manual_seed(1)


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (n, ci, h, co) in ((1, 512, 32, 512), (1, 512, 64, 512), (1, 256, 128, 256), (1, 128, 256, 128), (1, 256, 256, 256)):
    w = h
    x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(n, ci, generator=g)).to(DEV)
    d = (0.5 + torch.rand(n, co, generator=g)).to(DEV)
    noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    amax = H.absmax(x)
    aimg = H.split_activation(x, amax, in_scale=s)
    wp = H.pack_weight_fwd(wt)
    wimg = H.split_weight(wp, co, ci, 9)
    wpieces = H.split_weight_pieces(wp) if hasattr(H, 'split_weight_pieces') else None
    cls = H.classes_corr(h, w, 3, 3, 1)
    gf = 2.0 * n * h * w * ci * co * 9 / 1e9

    def v2(rows):
        out = H.empty_cl(n, co, h, w, DEV)
        H.conv_v2(aimg, wimg, out, cls, epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength, act='lrelu', alpha=0.2,
                  gain=1.4, clamp=-1.0, patch_rows=rows)
        return out

    def ig():
        out = H.empty_cl(n, co, h, w, DEV)
        H.conv_igemm(x, wp, ci, co, out, cls, in_scale=s, epi=L.EPI_FWD, out_scale=d, bias=bias, noise=noise, noise_nstride=0, noise_strength=strength,
                     act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, precision='f16x3', a_amax=amax, w_pieces=wpieces)
        return out

    line = f'{h}^2 x {ci}->{co} ({gf:.1f} GF):'
    ref = ig()
    t = timeit(ig)
    line += f'  igemm {t:6.1f} us ({gf / t * 1e3:5.0f} TF/s)'
    t = timeit(lambda: H.split_activation(x, amax, in_scale=s))
    line += f'  [split pass {t:4.1f} us]'
    for rows in (8, 4, 2):
        o = v2(rows)
        err = float((o - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: v2(rows))
        line += f'  v2/{rows} {t:6.1f} us ({gf / t * 1e3:5.0f} TF/s, diff {err:.1e})'
    print(line, flush=True)
