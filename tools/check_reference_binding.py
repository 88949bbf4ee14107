#!/usr/bin/env python3
"""Executes INTEGRATION.md section 1 against the real checkout.  BUILD CONTAINER ONLY (needs /root/reference; CPU is enough: nothing
here launches a kernel).  `tests/test_host.py::test_reference_binding_against_checkout` runs it in a subprocess when the checkout exists.

    python tools/check_reference_binding.py [/path/to/3DGAN-Inversion]

What it proves
  1. after `inv3d_amd.install_as_reference_modules()` the L1 / L2 module names of SURVEY.md section 8b resolve to this package, every
     other module of the checkout (training.warping_loss, torch_utils.misc / persistence, utils.camera_utils, configs ...) to the checkout;
  2. the reference's own `calc_warping_loss` imports and binds `G.synthesis` of this package's generator (called with a recording stub);
  3. every attribute chain the reference's callers apply to a generator object -- collected from the source of w_projector.project,
     BaseCoach / SingleIDCoach, calc_warping_loss, gen_interp_video, create_geometry, Space_Regulizer, log_utils -- exists on ours;
  4. `copy.deepcopy(G)`, `G.backbone.synthesis.named_buffers()`, `G.rendering_kwargs`, `.eval().float().requires_grad_()` behave;
  5. the pickle route: a generator built from the REFERENCE's classes and pickled with the reference's persistence machinery (module
     source embedded) unpickles -- through `persistence.import_hook` (persistence.py:149-177) -- into THIS package's classes with
     identical state dict, init kwargs, rendering kwargs and neural_rendering_resolution; `misc.copy_params_and_buffers` also works.
"""
import ast
import copy
import io
import os
import pickle
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
sys.path.insert(0, os.path.join(ROOT, '3dgan-inversion_amd'))
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self            # the checkout hard-codes .cuda() on small constants

RK = dict(depth_resolution=12, depth_resolution_importance=12, ray_start=2.25, ray_end=3.3, box_warp=1, disparity_space_sampling=False,
          clamp_mode='softplus', white_back=False, superresolution_module='training.superresolution.SuperresolutionHybrid8XDC',
          superresolution_noise_mode='none', sr_antialias=True, c_gen_conditioning_zero=False, c_scale=1, decoder_lr_mul=1,
          avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2], density_reg=0.25, density_reg_p_dist=0.004, reg_type='l1')
KW = dict(z_dim=64, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs={'num_layers': 2}, rendering_kwargs=RK,
          channel_base=2048, channel_max=16, fused_modconv_default='inference_only', num_fp16_res=0, sr_num_fp16_res=4,
          sr_kwargs={'channel_base': 2048, 'channel_max': 16, 'fused_modconv_default': 'inference_only'}, conv_clamp=None)


def reference_pickle() -> bytes:
    """A generator of the reference's own classes, pickled the way EG3D snapshots are (torch_utils/persistence.py)."""
    from training.triplane import TriPlaneGenerator
    assert os.path.realpath(sys.modules['training.triplane'].__file__).startswith(os.path.realpath(REF))
    torch.manual_seed(0)
    G = TriPlaneGenerator(**KW).eval().requires_grad_(False)
    G.neural_rendering_resolution = 128
    buf = io.BytesIO()
    pickle.dump(dict(G_ema=G), buf)
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    return buf.getvalue(), sd, G


def generator_attribute_chains():
    """Attribute chains applied to a generator object in the reference's callers: {'synthesis', 'backbone.synthesis.named_buffers', ...}."""
    files = ['training/projectors/w_projector.py', 'training/coaches/base_coach.py', 'training/coaches/single_id_coach.py',
             'training/warping_loss.py', 'gen_videos.py', 'criteria/localitly_regulizer.py', 'utils/log_utils.py']
    names = {'G', 'new_G', 'old_G', 'original_G'}
    chains = set()

    def chain(node):
        parts = []
        while isinstance(node, ast.Attribute):
            parts.append(node.attr)
            node = node.value
        if isinstance(node, ast.Name) and node.id in names:
            return list(reversed(parts))
        if isinstance(node, ast.Attribute):
            return None
        return None

    for rel in files:
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute):
                base = node
                while isinstance(base, ast.Attribute):
                    base = base.value
                ok = isinstance(base, ast.Name) and base.id in names
                if not ok and isinstance(base, ast.Name) and base.id == 'self':
                    # self.G.<chain> / self.original_G.<chain>
                    parts, n = [], node
                    while isinstance(n, ast.Attribute):
                        parts.append(n.attr)
                        n = n.value
                    parts.reverse()
                    if parts and parts[0] in names and len(parts) > 1:
                        chains.add('.'.join(parts[1:]))
                    continue
                if ok:
                    c = chain(node)
                    if c:
                        chains.add('.'.join(c))
    return chains


def main():
    blob, ref_sd, G_ref = reference_pickle()
    ref_kwargs = dict(G_ref.init_kwargs)
    # forget the checkout's hot-path modules, then bind
    for m in [m for m in sys.modules if m.split('.')[0] in ('training', 'torch_utils') and m not in ('torch_utils', 'training')]:
        if any(m == n or m.startswith(n + '.') for n in ('training.triplane', 'training.networks_stylegan2', 'training.superresolution',
                                                          'training.volumetric_rendering', 'torch_utils.ops')):
            del sys.modules[m]
    import inv3d_amd
    inv3d_amd.install_as_reference_modules()
    from inv3d_amd import reference_binding as RB

    # 1. names
    import importlib
    for name in RB.L1_MODULES + RB.L2_MODULES:
        mod = importlib.import_module(name)
        assert mod is importlib.import_module('inv3d_amd.' + name), name
        parent, leaf = name.rsplit('.', 1)
        assert getattr(importlib.import_module(parent), leaf) is mod, name
    from training.triplane import TriPlaneGenerator
    from torch_utils.ops import bias_act, upfirdn2d, conv2d_resample, conv2d_gradfix, filtered_lrelu, fma      # noqa: F401
    assert TriPlaneGenerator.__module__ == 'inv3d_amd.training.triplane'
    assert hasattr(bias_act, 'activation_funcs') and callable(upfirdn2d.setup_filter) and hasattr(conv2d_gradfix, 'no_weight_gradients')
    from torch_utils import misc, persistence
    from utils import camera_utils
    from configs import hyperparameters
    import training.warping_loss as WL
    for mod in (misc, persistence, camera_utils, hyperparameters, WL):
        assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REF)), mod.__file__
    print(f'1. {len(RB.L1_MODULES + RB.L2_MODULES)} hot-path modules bound; training.warping_loss, torch_utils.misc/persistence, utils, configs are the checkout\'s')

    # 2. the reference's calc_warping_loss binds our G.synthesis
    G = TriPlaneGenerator(**KW).eval().requires_grad_(False)
    G.neural_rendering_resolution = 128
    seen = {}

    class Stop(Exception):
        pass

    def spy(ws, c, **kw):
        seen.update(ws=tuple(ws.shape), c=tuple(c.shape), kw=kw)
        raise Stop
    G.synthesis = spy
    try:
        WL.calc_warping_loss(torch.zeros(1, 14, 512), torch.zeros(1, 25), None, None, None, None, None, G, None, None)
    except Stop:
        pass
    assert seen == dict(ws=(1, 14, 512), c=(1, 25), kw=dict(noise_mode='const', force_fp32=True)), seen
    del G.synthesis
    print('2. training.warping_loss.calc_warping_loss (checkout) -> G.synthesis(ws, c, noise_mode=\'const\', force_fp32=True) of this package')

    # 3. attribute chains used by the callers
    chains = generator_attribute_chains()
    missing = []
    for ch in sorted(chains):
        obj = G
        for part in ch.split('.'):
            if isinstance(obj, dict) or not hasattr(obj, part):
                if not (callable(obj) or isinstance(obj, dict)):       # attribute of a call result / dict item: beyond the generator object itself
                    missing.append(ch)
                break
            obj = getattr(obj, part)
    assert not missing, missing
    print(f'3. {len(chains)} generator attribute chains used by the reference callers resolve: {sorted(chains)}')

    # 4. object behaviour
    nb = [n for n, _ in G.backbone.synthesis.named_buffers() if 'noise_const' in n]
    nb2 = [n for n, _ in G.superresolution.named_buffers() if 'noise_const' in n]
    assert len(nb) == 13 and len(nb2) == 4, (nb, nb2)
    G2 = copy.deepcopy(G).eval().requires_grad_(False).float()
    assert type(G2) is type(G) and all(torch.equal(a, b) for a, b in zip(G.state_dict().values(), G2.state_dict().values()))
    assert G2.rendering_kwargs == RK and G2.neural_rendering_resolution == 128 and G.z_dim == 64
    assert set(G.state_dict()) == set(ref_sd), set(G.state_dict()) ^ set(ref_sd)
    print(f'4. deepcopy / named_buffers ({len(nb)} + {len(nb2)} noise_const) / rendering_kwargs / state-dict keys ({len(ref_sd)}) as the reference class')

    # 5. pickle route
    G_un = pickle.loads(blob)['G_ema']
    assert type(G_un).__name__ == 'TriPlaneGenerator' and isinstance(G_un, TriPlaneGenerator), type(G_un).__mro__
    for sub in (G_un.backbone, G_un.backbone.synthesis, G_un.backbone.synthesis.b64.conv0, G_un.backbone.mapping.fc0, G_un.superresolution,
                G_un.superresolution.block1.torgb, G_un.decoder, G_un.decoder.net[0], G_un.renderer, G_un.ray_sampler):
        assert any(c.__module__.startswith('inv3d_amd.') for c in type(sub).__mro__), (type(sub).__mro__,)
        assert not any(c.__module__.startswith(('training.', '_imported_module_')) for c in type(sub).__mro__), type(sub).__mro__
    sd = G_un.state_dict()
    assert set(sd) == set(ref_sd)
    assert all(torch.equal(sd[k], ref_sd[k]) for k in ref_sd)
    assert G_un.neural_rendering_resolution == 128 and dict(G_un.rendering_kwargs) == RK and not G_un.training
    assert dict(G_un.init_kwargs) == ref_kwargs
    G3 = TriPlaneGenerator(**G_un.init_kwargs).eval().requires_grad_(False)          # gen_samples.py:146-152 pattern
    misc.copy_params_and_buffers(G_un, G3, require_all=True)
    assert all(torch.equal(a, b) for a, b in zip(G3.state_dict().values(), G_un.state_dict().values()))
    blob2 = pickle.dumps(dict(G_ema=G_un))                                   # and back out through the reference's persistence
    G4 = pickle.loads(blob2)['G_ema']
    assert all(torch.equal(a, b) for a, b in zip(G4.state_dict().values(), G_un.state_dict().values()))
    print(f'5. reference pickle ({len(blob) / 1e6:.1f} MB, embedded NVIDIA source) -> inv3d_amd classes via persistence.import_hook: '
          f'{len(sd)} tensors identical; copy_params_and_buffers and re-pickling work')
    print('REFERENCE BINDING OK')


if __name__ == '__main__':
    main()
