"""Binding a 3DGAN-Inversion checkout to the MI355X path.

    import inv3d_amd
    inv3d_amd.install_as_reference_modules()        # after the checkout is on sys.path, before `from training...` imports

replaces, in `sys.modules`, exactly the modules that make up the hot path

    torch_utils.ops.{bias_act, upfirdn2d, filtered_lrelu, conv2d_resample, conv2d_gradfix, fma}                      (L1 operators)
    training.{networks_stylegan2, superresolution, triplane}                                                          (L2 model)
    training.volumetric_rendering.{renderer, ray_sampler, math_utils}

by this package's modules of the same names (the same module OBJECTS, so module-level state such as
`conv2d_gradfix.no_weight_gradients` is shared).  Everything else keeps resolving to the checkout: training.coaches,
training.projectors, training.warping_loss, torch_utils.misc / persistence / custom_ops, dnnlib, utils, configs ...  So
`scripts/run_pti.py`, the projector and the coaches run unchanged (`from training.triplane import TriPlaneGenerator` is the class
of this package; `calc_warping_loss(..., G, ...)` calls its `G.synthesis`).

EG3D pickles rebuild their classes from module source embedded in the file (torch_utils/persistence.py:181-204): only the names
that source IMPORTS resolve locally, i.e. L1 is swapped by the aliasing alone, the L2 classes are not.  `persistence_import_hook`
(registered by install_as_reference_modules through the reference's own `persistence.import_hook`, :149-177) redirects the embedded
module of every L2 class to this package, and `ReferenceStateMixin.__setstate__` rebuilds the object with this package's
constructor from the pickled constructor arguments and adopts the pickled parameters / buffers / children.
`utils/models_utils.load_old_G()` then returns a generator that runs on the gfx950 kernels.
"""
import importlib
import sys
from typing import Dict, Optional

L1_MODULES = ('torch_utils.ops.bias_act', 'torch_utils.ops.upfirdn2d', 'torch_utils.ops.filtered_lrelu', 'torch_utils.ops.conv2d_resample',
              'torch_utils.ops.conv2d_gradfix', 'torch_utils.ops.fma')
L2_MODULES = ('training.networks_stylegan2', 'training.superresolution', 'training.triplane', 'training.volumetric_rendering.renderer',
              'training.volumetric_rendering.ray_sampler', 'training.volumetric_rendering.math_utils')

# class name -> module of this package that defines it (what a pickle's embedded source is redirected to)
L2_CLASSES: Dict[str, str] = {
    'FullyConnectedLayer': 'training.networks_stylegan2', 'MappingNetwork': 'training.networks_stylegan2',
    'SynthesisLayer': 'training.networks_stylegan2', 'ToRGBLayer': 'training.networks_stylegan2', 'SynthesisBlock': 'training.networks_stylegan2',
    'SynthesisNetwork': 'training.networks_stylegan2', 'Generator': 'training.networks_stylegan2',
    'SuperresolutionHybrid8XDC': 'training.superresolution', 'TriPlaneGenerator': 'training.triplane', 'OSGDecoder': 'training.triplane',
}

_installed = False


def _alias(name: str) -> None:
    mod = importlib.import_module('inv3d_amd.' + name)
    parent_name, _, leaf = name.rpartition('.')
    try:
        parent = importlib.import_module(parent_name)                 # the CHECKOUT's package (training, torch_utils.ops, ...)
    except ImportError as e:
        raise ImportError(f'install_as_reference_modules: cannot import the reference package {parent_name!r}; put the 3DGAN-Inversion '
                          f'checkout on sys.path first (or pass reference_root=...)') from e
    if parent.__name__.startswith('inv3d_amd'):
        raise ImportError(f'{parent_name!r} resolves to inv3d_amd itself: do NOT put .../inv3d_amd on sys.path, put the checkout there')
    sys.modules[name] = mod
    setattr(parent, leaf, mod)


def persistence_import_hook(meta):
    """torch_utils.persistence import hook (persistence.py:149-177): an L2 class being unpickled is taken from this package instead of
    from the source embedded in the pickle.  The embedded source is replaced by a one-line module that imports the class."""
    target = L2_CLASSES.get(meta.class_name)
    if target is not None and f'class {meta.class_name}(' in meta.module_src:
        meta.module_src = f'from inv3d_amd.{target} import {meta.class_name}  # redirected by inv3d_amd.reference_binding\n'
    return meta


def install_as_reference_modules(reference_root: Optional[str] = None, hook_pickles: bool = True) -> None:
    """See the module docstring.  Idempotent.  `reference_root`: path of the checkout, appended to sys.path if given."""
    global _installed
    if reference_root is not None and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    for name in L1_MODULES + L2_MODULES:
        _alias(name)
    if hook_pickles and not _installed:
        persistence = importlib.import_module('torch_utils.persistence')      # the checkout's own
        persistence.import_hook(persistence_import_hook)
    _installed = True


class ReferenceStateMixin:
    """Mixed into the L2 module classes.  `__setstate__` with the state of a REFERENCE object (an unpickled EG3D generator: the
    reference class's `__dict__`, recognisable by `_init_args` / `_init_kwargs` recorded by persistence.persistent_class and by the
    absence of this package's marker) rebuilds the object with this package's constructor and adopts the pickled tensors and children;
    any other state (copy.deepcopy, torch.save of this package's own objects) takes torch.nn.Module's default path."""
    _inv3d_native = True

    def __setstate__(self, state):
        import torch
        if '_inv3d_marker' in state or '_init_kwargs' not in state:
            torch.nn.Module.__setstate__(self, state)
            return
        args, kwargs = tuple(state.get('_init_args', ())), dict(state.get('_init_kwargs', {}))
        self.__init__(*args, **kwargs)
        for k, v in state.get('_parameters', {}).items():
            if v is not None:
                if k not in self._parameters:
                    raise KeyError(f'{type(self).__name__}: pickled parameter {k!r} has no counterpart')
                self._parameters[k] = v
        for k, v in state.get('_buffers', {}).items():
            if v is not None:
                self._buffers[k] = v
        for k, child in state.get('_modules', {}).items():
            if k not in self._modules:
                raise KeyError(f'{type(self).__name__}: pickled child module {k!r} has no counterpart')
            # children were unpickled before their parent: L2 classes are already this package's; parameter-free helpers of the
            # reference (its ImportanceRenderer / RaySampler objects) are dropped in favour of the ones the constructor just built
            if isinstance(child, ReferenceStateMixin) or any(True for _ in child.parameters()):
                self._modules[k] = child
        self.training = bool(state.get('training', False))
        for k in ('neural_rendering_resolution', 'rendering_kwargs'):          # plain attributes callers read back (triplane.py:44-45)
            if k in state:
                setattr(self, k, state[k])

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_inv3d_marker'] = True
        return state
