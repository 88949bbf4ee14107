"""Source-free generator archives.

EG3D checkpoints are pickles of `@persistence.persistent_class` objects: unpickling re-executes the module source embedded in the
file (torch_utils/persistence.py:120-128,181-204; loaders utils/models_utils.py:21-25, legacy.py:24-60).  The MI355X path does not
unpickle code.  `tools/convert_eg3d_pickle.py` -- run once on a machine that has the reference tree and the pickle -- writes a flat
archive: every tensor of `G_ema.state_dict()` (names as in SURVEY.md Appendix C) in safetensors format, plus the constructor
arguments (`init_kwargs`), `rendering_kwargs` and `neural_rendering_resolution` as JSON metadata.  `load_generator` rebuilds the
module tree of this package from that metadata and loads the tensors strictly (a missing or unexpected key is an error)."""
import json
from typing import Dict, Tuple

import torch

FORMAT = 'inv3d_amd.generator.v1'


def _plain(o):
    """EasyDict / tuple / numpy scalars -> JSON-serialisable plain python."""
    if isinstance(o, dict):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if hasattr(o, 'item') and not isinstance(o, (str, bytes)):
        try:
            return o.item()
        except Exception:
            pass
    return o


def save_generator_archive(path: str, state_dict: Dict[str, torch.Tensor], init_kwargs: dict, neural_rendering_resolution: int) -> None:
    from safetensors.torch import save_file
    tensors = {k: v.detach().to('cpu').contiguous().clone() for k, v in state_dict.items()}
    meta = dict(format=FORMAT, init_kwargs=json.dumps(_plain(init_kwargs)), neural_rendering_resolution=str(int(neural_rendering_resolution)))
    save_file(tensors, path, metadata=meta)


def read_generator_archive(path: str) -> Tuple[Dict[str, torch.Tensor], dict, int]:
    from safetensors import safe_open
    tensors = {}
    with safe_open(path, framework='pt', device='cpu') as f:
        meta = f.metadata() or {}
        if meta.get('format') != FORMAT:
            raise ValueError(f'{path}: not a {FORMAT} archive (metadata format = {meta.get("format")!r})')
        for k in f.keys():
            tensors[k] = f.get_tensor(k)
    return tensors, json.loads(meta['init_kwargs']), int(meta['neural_rendering_resolution'])


def load_generator(path: str, device='cuda'):
    """Archive -> TriPlaneGenerator of this package on `device`, eval mode, fp32, gradients off (as utils/models_utils.load_old_G)."""
    from .training.triplane import TriPlaneGenerator
    tensors, kw, nrr = read_generator_archive(path)
    G = TriPlaneGenerator(**kw)
    missing, unexpected = G.load_state_dict(tensors, strict=False)
    if missing or unexpected:
        raise KeyError(f'{path}: state-dict mismatch; missing {sorted(missing)[:8]}, unexpected {sorted(unexpected)[:8]}')
    G.neural_rendering_resolution = nrr
    return G.eval().float().requires_grad_(False).to(device)
