#!/usr/bin/env python3
"""EG3D pickle -> source-free archive for inv3d_amd.weights.load_generator.

Run ONCE on a machine that has the reference tree (its dnnlib / torch_utils are needed to unpickle: the pickle re-creates its classes
from embedded source, torch_utils/persistence.py:181-204) and the checkpoint; the output contains tensors and JSON only.

    python tools/convert_eg3d_pickle.py --reference /path/to/3DGAN-Inversion --pkl ffhqrebalanced512-128.pkl --out ffhq512-128.safetensors
    python tools/convert_eg3d_pickle.py --reference ... --pkl model_<id>_<type>.pt --torch-load --out tuned.safetensors   # PTI output
"""
import argparse
import os
import pickle
import sys


def convert(reference: str, pkl: str, out: str, key: str = 'G_ema', torch_load: bool = False) -> None:
    sys.path.insert(0, os.path.abspath(reference))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), '3dgan-inversion_amd'))
    import torch
    from inv3d_amd import weights
    with open(pkl, 'rb') as f:
        if torch_load:                       # utils/models_utils.load_tuned_G: torch.save of the whole module
            G = torch.load(f, map_location='cpu', weights_only=False)
        else:                                # utils/models_utils.load_old_G
            data = pickle.load(f)
            G = data[key] if isinstance(data, dict) else data
    G = G.eval().float().cpu()
    weights.save_generator_archive(out, G.state_dict(), dict(G.init_kwargs), getattr(G, 'neural_rendering_resolution', 64))
    n = sum(v.numel() for v in G.state_dict().values())
    print(f'wrote {out}: {len(G.state_dict())} tensors, {n / 1e6:.2f} M values')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', required=True, help='root of the cvlab-kaist/3DGAN-Inversion (or EG3D) source tree')
    ap.add_argument('--pkl', required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--key', default='G_ema')
    ap.add_argument('--torch-load', action='store_true')
    a = ap.parse_args()
    convert(a.reference, a.pkl, a.out, a.key, a.torch_load)
