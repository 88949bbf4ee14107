"""Forward-only tri-plane decode (gather + 32->64->33 MLP on the fp32 matrix cores) over as many points as one render pass samples."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
from inv3d_amd import synthetic as S
from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
from inv3d_amd.training.triplane import OSGDecoder
dev = 'cuda'
torch.manual_seed(0)
planes = (torch.randn(1, 96, 256, 256, device=dev) * 0.5).contiguous(memory_format=torch.channels_last)
dec = OSGDecoder(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32}).to(dev)
opts = S.default_rendering_kwargs()
R = ImportanceRenderer()
for M in (786432, 1572864):
    coords = (torch.rand(1, M, 3, device=dev) - 0.5) * 0.9
    dirs = torch.zeros_like(coords)
    with torch.no_grad():
        for _ in range(3): R.run_model(planes.view(1, 3, 32, 256, 256), dec, coords, dirs, opts)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): R.run_model(planes.view(1, 3, 32, 256, 256), dec, coords, dirs, opts)
        torch.cuda.synchronize()
    print(f'M={M}: {(time.perf_counter()-t)*100:.3f} ms per decode')
