"""Weight import: reference pickle -> source-free archive -> this package's generator.
Needs the reference tree to build (and unpickle) a checkpoint, so it only runs in the build container."""
import io
import os
import pickle
import subprocess
import sys

import pytest
import torch

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_archive_round_trip(tmp_path):
    from inv3d_amd import synthetic as S, weights as W
    G = S.make_generator(w_dim=32, z_dim=32, plane_res=32, channel_base=256, channel_max=16, nrr=16, sr_in_res=16, sr_widths=(16, 8), device='cpu')
    S.load_synthetic_weights(G, 0)
    kw = dict(z_dim=32, c_dim=25, w_dim=32, img_resolution=64, img_channels=3, sr_num_fp16_res=4, mapping_kwargs={'num_layers': 2},
              rendering_kwargs=G.rendering_kwargs, sr_kwargs={'channel_base': 256, 'channel_max': 16, 'fused_modconv_default': 'inference_only',
                                                              'sr_widths': (16, 8), 'input_resolution': 16, 'w_dim': 32},
              plane_resolution=32, channel_base=256, channel_max=16, fused_modconv_default='inference_only', conv_clamp=None)
    p = str(tmp_path / 'g.safetensors')
    W.save_generator_archive(p, G.state_dict(), kw, 16)
    G2 = W.load_generator(p, device='cpu')
    assert G2.neural_rendering_resolution == 16
    for (k, a), (k2, b) in zip(G.state_dict().items(), G2.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    bad = str(tmp_path / 'bad.safetensors')
    sd = dict(G.state_dict())
    sd.pop('decoder.net.0.weight')
    W.save_generator_archive(bad, sd, kw, 16)
    with pytest.raises(KeyError):
        W.load_generator(bad, device='cpu')


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_convert_reference_pickle(tmp_path):
    """A checkpoint pickled by the reference's own persistence machinery (embedded source and all) goes through
    tools/convert_eg3d_pickle.py in a subprocess and loads into this package's module tree with identical tensors."""
    make = f'''
import sys, pickle, torch
sys.path.insert(0, {REF!r})
import dnnlib
from training.triplane import TriPlaneGenerator
torch.manual_seed(0)
rk = dict(superresolution_module='training.superresolution.SuperresolutionHybrid8XDC', sr_antialias=True, c_gen_conditioning_zero=False,
          c_scale=1.0, decoder_lr_mul=1.0, depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
          superresolution_noise_mode='none', avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2], disparity_space_sampling=False,
          clamp_mode='softplus', white_back=False)
G = TriPlaneGenerator(z_dim=64, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=4, mapping_kwargs=dnnlib.EasyDict(num_layers=2),
                      rendering_kwargs=rk, sr_kwargs=dnnlib.EasyDict(channel_base=512, channel_max=32, fused_modconv_default='inference_only'),
                      channel_base=512, channel_max=32, fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None).eval()
G.neural_rendering_resolution = 128
with torch.no_grad():
    for p in G.parameters():
        p.copy_(torch.randn_like(p))
pickle.dump(dict(G_ema=G), open({str(tmp_path / "ref.pkl")!r}, 'wb'))
torch.save(G.state_dict(), {str(tmp_path / "ref_sd.pt")!r})
'''
    subprocess.run([sys.executable, '-c', make], check=True, timeout=600)
    out = str(tmp_path / 'conv.safetensors')
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'convert_eg3d_pickle.py'), '--reference', REF, '--pkl', str(tmp_path / 'ref.pkl'),
                    '--out', out], check=True, timeout=600)
    from inv3d_amd import weights as W
    G = W.load_generator(out, device='cpu')
    ref_sd = torch.load(str(tmp_path / 'ref_sd.pt'))
    sd = G.state_dict()
    assert set(sd) == set(ref_sd)
    for k, v in ref_sd.items():
        assert torch.equal(sd[k], v.float()), k
    assert G.neural_rendering_resolution == 128 and G.rendering_kwargs['box_warp'] == 1
    assert 'torch_utils' not in sys.modules or not getattr(sys.modules['torch_utils'], '__file__', '').startswith(REF)
