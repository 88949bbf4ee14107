"""List-length statistics of the row-keyed plane-gradient scatter (EG3D_SCATTER=4): how the (tile, texel row) lists are balanced."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import synthetic as S, _lib as L, hipops as H
from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
from inv3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
from inv3d_amd.training.triplane import OSGDecoder
torch.manual_seed(0)
dev = 'cuda'
planes = (torch.randn(1, 96, 256, 256, device=dev) * 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
dec = OSGDecoder(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32}).to(dev)
cam = S.synth_cameras(1).to(dev)
c2w = cam[:, :16].reshape(1, 4, 4); K = cam[:, 16:].reshape(1, 3, 3)
o, d = RaySampler()(c2w, K, 128)
orig = L.check
def check(rc, name):
    orig(rc, name)
    if name == 'triplane_scatter':
        ws = sys._getframe(1).f_locals['ws']
        torch.cuda.synchronize()
        nb16 = 3 * 18 * 18 * 16
        counts = ws[:nb16].cpu().view(3, 324, 16).long()
        tot = counts.sum(-1)
        print('pairs', int(counts.sum()), 'busy tiles', int((tot > 0).sum()), 'of', tot.numel())
        for pl in range(3):
            t = tot[pl]; c = counts[pl]
            mx = c.max(-1).values
            print(f'plane {pl}: tile total mean {t[t>0].float().mean():.0f} max {int(t.max())}; per-tile longest list mean {mx[t>0].float().mean():.0f} max {int(mx.max())};'
                  f' sum over tiles of longest list {int(mx.sum())} vs pairs/16 {int(t.sum()) // 16}')
        mx = counts.max(-1).values.flatten()
        srt, _ = mx.sort(descending=True)
        print('longest lists of the 20 heaviest tiles', srt[:20].tolist())
L.check = check; H.L.check = check
rgb, dep, w = ImportanceRenderer()(planes, dec, o, d, S.default_rendering_kwargs())
(rgb.sum() + dep.sum()).backward()
