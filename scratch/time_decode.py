import sys, torch, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'3dgan-inversion_amd')
from inv3d_amd import synthetic as S
from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
from inv3d_amd.training.triplane import OSGDecoder
dev='cuda'; torch.manual_seed(0)
planes = (torch.randn(1,96,256,256,device=dev)*0.5).contiguous(memory_format=torch.channels_last)
dec = OSGDecoder(32, {'decoder_lr_mul':1.0,'decoder_output_dim':32}).to(dev)
coords = (torch.rand(1, 1572864, 3, device=dev)-0.5)*0.9
R = ImportanceRenderer(); opts = S.default_rendering_kwargs()
for _ in range(3): out = R.run_model(planes, dec, coords, None, opts)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): out = R.run_model(planes, dec, coords, None, opts)
torch.cuda.synchronize(); print('VALU' if os.environ.get('EG3D_VALU_DECODE') else 'MFMA', 'decode 1.57M pts ms', (time.perf_counter()-t)*50, float(out['rgb'].sum()), float(out['sigma'].sum()))
