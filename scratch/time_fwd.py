import sys, torch, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'3dgan-inversion_amd')
from inv3d_amd import synthetic as S
from inv3d_amd.training.volumetric_rendering.renderer import ImportanceRenderer
from inv3d_amd.training.volumetric_rendering.ray_sampler import RaySampler
from inv3d_amd.training.triplane import OSGDecoder
dev='cuda'; torch.manual_seed(0); N=1
planes = (torch.randn(N,96,256,256,device=dev)*0.5).contiguous(memory_format=torch.channels_last)
dec = OSGDecoder(32, {'decoder_lr_mul':1.0,'decoder_output_dim':32}).to(dev)
cam = S.synth_cameras(N).to(dev); c2w = cam[:,:16].reshape(N,4,4); K = cam[:,16:].reshape(N,3,3)
opts = S.default_rendering_kwargs(); R = ImportanceRenderer(); rs = RaySampler()
def run():
    with torch.no_grad():
        o,d = rs(c2w, K, 128); rgb, dep, ws = R(planes, dec, o, d, opts)
for _ in range(3): run()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); print('DBG', os.environ.get('EG3D_DBG'), 'fwd ms', (time.perf_counter()-t)*50)
