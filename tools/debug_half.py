import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import hipops as H, _lib as L
DEV = 'cuda'
g = torch.Generator().manual_seed(1)
n, ci, h, w, co = 1, 128, 256, 256, 128
x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
s = (1 + 0.5 * torch.randn(n, ci, generator=g)).to(DEV)
d = (0.5 + torch.rand(n, co, generator=g)).to(DEV)
noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
aimg = H.split_activation(x, H.absmax(x), in_scale=s)
wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
cls = H.classes_corr(h, w, 3, 3, 1)
z = torch.nn.functional.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1) * d.double()[:, :, None, None]
ref = (torch.nn.functional.leaky_relu(z + noise.double() * 0.3 + bias.double()[None, :, None, None], 0.2) * 1.4).float()
def fwd(rows, use_noise=True, use_bias=True, use_d=True, amax=True):
    out = H.empty_cl(n, co, h, w, DEV); am = torch.zeros(1, device=DEV)
    H.conv_v2(aimg, wimg, out, cls, epi=L.EPI_FWD, out_scale=d if use_d else None, bias=bias if use_bias else None, noise=noise if use_noise else None, noise_nstride=0,
              noise_strength=strength if use_noise else None, act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0, out_amax=am if amax else None, patch_rows=rows)
    torch.cuda.synchronize()
    return out
for rows in (8, 4):
    errs = [float((fwd(rows) - ref).abs().max()) for _ in range(6)]
    print('rows', rows, 'max err vs torch', ['%.2e' % e for e in errs])
for kw in (dict(use_noise=False), dict(use_bias=False), dict(use_d=False), dict(amax=False), dict(use_noise=False, use_bias=False, use_d=False, amax=False)):
    a = fwd(8, **kw)
    bad = sum(int((fwd(4, **kw) != a).any()) for _ in range(8))
    print(kw, 'rows4 != rows8 in', bad, '/ 8')
out = fwd(4); dd = (out - ref).abs(); idx = (dd > 1e-3).nonzero()
print('bad elems', len(idx), 'channels', sorted(set(idx[:, 1].tolist()))[:40], 'rows%4', sorted(set((idx[:, 2] % 4).tolist())), 'cols%32', sorted(set((idx[:, 3] % 32).tolist()))[:40])
print('---- detail')
zc = (z + bias.double()[None, :, None, None]).float()      # pre-activation without noise
for k in range(min(6, len(idx))):
    _, c, y, xx = idx[k].tolist()
    got = float(out[0, c, y, xx]); want = float(ref[0, c, y, xx])
    pre = float(zc[0, c, y, xx]); nzv = float(noise[y, xx]) * 0.3
    # invert: which noise value would produce `got`?
    g = got / 1.4
    pre_got = g if g > 0 else g / 0.2
    print(f'c {c} y {y} x {xx}: got {got:.5f} want {want:.5f}; pre(no noise) {pre:.5f} noise*s {nzv:.5f}; implied noise term {pre_got - pre:.5f};',
          'neighbours', [round(float(noise[(y + dy) % 256, (xx + dx) % 256]) * 0.3, 5) for dy, dx in ((0, -1), (0, 1), (-1, 0), (1, 0), (0, 16), (0, -16), (4, 0), (-4, 0))])
