"""HIP-graph replay behind the plain `G.synthesis(ws, c, **kw)` call (the reference's drop-in callers never see a graph object).

Why: one forward + backward of the full-size generator is ~190 kernel launches.  Launched one by one through Python / ctypes they cost
~20 us of host time each, i.e. ~16 ms per optimisation step for ~6 ms of GPU work -- the loops of the reference
(training/projectors/w_projector.py:189-261, training/coaches/base_coach.py:162-164 + single_id_coach.py:64-77) call `G.synthesis` exactly
like that.  `inversion.LatentProjector(use_graph=True)` removes the cost by capturing its whole step, but a caller bound through
`reference_binding.install_as_reference_modules()` runs the reference's own loop.  So the generator does it itself:

  * a call signature (shapes, keyword arguments, grad mode, which leaves require grad, versions of the frozen weights) that has been seen
    `HOT_AFTER` times is captured once -- forward into one `hipGraph`, `torch.autograd.grad` of its outputs into a second one sharing the
    memory pool -- and every later call with that signature is: copy (ws, c) into the static inputs, replay, hand back copies of the
    outputs, wrapped in ONE autograd node whose backward copies the output gradients in, replays the second graph and returns the static
    input / parameter gradients;
  * anything the capture cannot express falls back to the per-launch path, silently and correctly: a call made while the previous
    forward's graph is still waiting for its backward (two live graphs of one generator), `noise_inject`, cached-backbone calls, a
    launch profiler, a call that is itself being captured (LatentProjector / PivotalTuner graphs), CPU tensors.

Semantics kept: outputs are fresh tensors (copies of the static buffers); gradients of `ws` / `c` flow on into the caller's graph; parameter
gradients reach `.grad` through autograd's own AccumulateGrad (they alias static buffers when `.grad` was None, which is what
`torch.cuda.make_graphed_callables` does too; a `.grad` that still aliases the buffer from the previous step -- `zero_grad(set_to_none=
False)` -- gets a private copy first).  Trainable weights are re-packed inside the captured forward, so optimiser steps between calls are
seen; frozen weights are part of the signature (pointer + version), so `load_state_dict` / in-place edits re-capture.
Set `EG3D_GRAPH_EAGER=0` (or `inv3d_amd.graphed.ENABLED = False`, or `G.graph_eager = False`) to switch it off.
"""
import os
import weakref

import torch

from . import hipops as H

ENABLED = os.environ.get('EG3D_GRAPH_EAGER', '1') != '0'
HOT_AFTER = 2      # per-launch calls of a signature before it is captured (they double as warm-up)
MAX_ENTRIES = 3                                                     # captured signatures kept per generator (each holds its activations)

_STATE = weakref.WeakKeyDictionary()        # generator -> _PerG   (not an attribute: copy.deepcopy(G) must not meet graph objects)
STATS = dict(captured=0, replayed=0, eager=0, fallback_pending=0, capture_failed=0)


class _PerG:
    def __init__(self):
        self.counts, self.entries, self.failed = {}, {}, set()


class _Entry:
    pass


def _leaves(G):
    """(leaf tensors that require grad, signature of everything frozen, signature of the trainable PARAMETERS).  Parameters and buffers in
    module order.  Frozen tensors enter with (pointer, in-place version): `load_state_dict` / in-place edits re-capture (a raw `.data`
    write bumps no version: call `graphed.reset(G)` after one).  Trainable ones enter with their identity and pointer only -- optimiser
    steps between calls must NOT re-capture; their derived images are rebuilt inside every replay instead (`pack_inside`)."""
    leaves, frozen, trainable = [], [], []
    for t in G.parameters():
        if t.requires_grad:
            leaves.append(t)
            trainable.append((id(t), t.data_ptr()))
        else:
            frozen.append((t.data_ptr(), t._version))
    for t in G.buffers():
        if t.requires_grad:
            leaves.append(t)
        else:
            frozen.append((t.data_ptr(), t._version))
    return leaves, hash(tuple(frozen)), tuple(trainable)


def _render_key(G):
    """Hashable snapshot of G.rendering_kwargs (depth resolutions, ray range, box_warp, white_back ... are baked into a captured graph; the
    reference's viewers mutate them between calls: viz/renderer.py, gen_videos.py).  None = not hashable: no replay for this call."""
    rk = getattr(G, 'rendering_kwargs', None) or {}
    out = []
    for k in sorted(rk):
        v = rk[k]
        if isinstance(v, (list, tuple)) and all(isinstance(x, (bool, int, float, str, type(None))) for x in v):
            v = tuple(v)
        elif not isinstance(v, (bool, int, float, str, type(None))):
            return None
        out.append((k, v))
    return tuple(out)


def _kw_key(kw):
    out = []
    for k in sorted(kw):
        v = kw[k]
        if isinstance(v, (bool, int, float, str, type(None))):
            out.append((k, v))
        else:
            return None              # a tensor / dict valued keyword: not a signature we can replay
    return tuple(out)


class _GraphedFn(torch.autograd.Function):
    """inputs: (entry, ws, c, u1 | None, u2 | None, *leaves) -> copies of the entry's static outputs."""

    @staticmethod
    def forward(ctx, entry, ws, c, u1, u2, *leaves):
        e = entry
        e.s_ws.copy_(ws)
        e.s_c.copy_(c)
        if e.s_u1 is not None:
            e.s_u1.copy_(u1)
            if e.s_u2 is not None:
                e.s_u2.copy_(u2)
        e.fwd.replay()
        e.gen += 1
        ctx.entry, ctx.gen = e, e.gen
        e.pending = weakref.ref(ctx)
        e.done = False
        outs = tuple(o.clone(memory_format=torch.preserve_format) for o in e.s_out)
        ctx.mark_non_differentiable(*[o for o, r in zip(outs, e.out_req) if not r])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable       # a replayed backward has no graph of its own: double backward raises
    def backward(ctx, *gouts):
        e = ctx.entry
        if ctx.gen != e.gen:
            raise RuntimeError('G.synthesis: backward through a forward whose captured activations were overwritten by a later call '
                               '(set EG3D_GRAPH_EAGER=0 to keep several forwards of one signature alive)')
        gi = 0
        for g, r in zip(gouts, e.out_req):
            if not r:
                continue
            if g is None:
                e.s_gout[gi].zero_()
            else:
                e.s_gout[gi].copy_(g)
            gi += 1
        # a parameter whose .grad still aliases our static buffer (zero_grad(set_to_none=False)) must not be accumulated onto itself
        for t, s in zip(e.grad_targets, e.s_gin):
            if s is not None and t.is_leaf and t.grad is not None and t.grad.data_ptr() == s.data_ptr():
                t.grad = t.grad.clone()
        e.bwd.replay()
        e.done = True
        res = [None, None, None, None, None]        # entry, ws, c, u1, u2
        k = 0
        if e.ws_req:
            res[1] = e.s_gin[k].detach() if e.s_gin[k] is not None else None
            k += 1
        if e.c_req:
            res[2] = e.s_gin[k].detach() if e.s_gin[k] is not None else None
            k += 1
        res += [s.detach() if s is not None else None for s in e.s_gin[k:]]
        return tuple(res)


class _aliased_leaves:
    """For the duration of a capture, every leaf of the generator that requires grad (trainable parameters, the noise_const buffers of a
    latent projection) is replaced inside its module by a fresh leaf sharing the same storage.  The captured autograd graph is then
    entirely private: it never touches the caller's tensors' gradient accumulators -- those were created on whatever stream the caller's
    earlier (per-launch) iterations ran on, usually the default stream, and autograd synchronises a backward with its accumulators'
    streams: a cross-stream wait on a non-capturing stream inside hipStreamBeginCapture ... EndCapture (observed: a segfault in
    capture_end).  `aliases` is aligned with `leaves`."""

    def __init__(self, G, leaves):
        self.G, self.leaves = G, leaves
        self.aliases, self.undo = [], []

    def __enter__(self):
        by_id = {}
        for t in self.leaves:
            a = t.detach()
            a = torch.nn.Parameter(a, requires_grad=True) if isinstance(t, torch.nn.Parameter) else a.requires_grad_(True)
            by_id[id(t)] = a
            self.aliases.append(a)
        if by_id:
            for m in self.G.modules():
                for store in (m._parameters, m._buffers):
                    for name, t in store.items():
                        if t is not None and id(t) in by_id:
                            self.undo.append((store, name, t))
                            store[name] = by_id[id(t)]
        return self.aliases

    def __exit__(self, *exc):
        for store, name, t in self.undo:
            store[name] = t
        return False


def _capture(G, impl, ws, c, uni, kw, leaves, need_grad, pack_inside):
    """Capture forward (and backward) for this signature; returns the entry.  Two passes: the first sizes the zero arenas (every
    zero-initialised accumulator of the step comes out of one buffer cleared by one launch, hipops.ZeroArena), the second is kept."""
    dev = ws.device
    e = _Entry()
    e.ws_req, e.c_req = bool(need_grad and ws.requires_grad), bool(need_grad and c.requires_grad)
    e.s_ws = ws.detach().clone().requires_grad_(e.ws_req)
    e.s_c = c.detach().clone().requires_grad_(e.c_req)
    e.s_u1 = uni[0].detach().clone() if uni is not None else None
    e.s_u2 = uni[1].detach().clone() if (uni is not None and uni[1] is not None) else None
    e.arena_f, e.arena_b = H.ZeroArena(dev), H.ZeroArena(dev)
    e.gen, e.pending, e.done = 0, None, True
    # trainable weights (whether or not THIS call differentiates: a no-grad preview between optimiser steps replays too): every derived
    # image must be rebuilt INSIDE the captured forward -- a new optimiser epoch makes the caches of trainable weights miss (frozen ones keep
    # their images: fused.WeightCache.key_of, hipops.memo)
    s_uni = (e.s_u1, e.s_u2) if e.s_u1 is not None else None
    for attempt in range(2):
        keep = attempt == 1
        if pack_inside:
            H.weights_changed()
        refs = H.CAPTURE_REFS = []     # every derived tensor the launches of this capture were handed (they hold raw pointers to them)
        try:
            e_fwd_bwd = _capture_once(G, impl, e, kw, leaves, need_grad, s_uni)
        finally:
            H.CAPTURE_REFS = None
        fwd, bwd, outs, req, gouts, gins, padded = e_fwd_bwd
        if keep:
            e.fwd, e.bwd = fwd, bwd
            e.s_out = [o.detach() for o in outs]
            e.out_req, e.padded = req, padded
            e.s_gout = gouts
            e.s_gin = list(gins) if gins is not None else []
            e.grad_targets = ([ws] if e.ws_req else []) + ([c] if e.c_req else []) + list(leaves)      # .grad holders (the user's tensors)
            e.refs = refs
        del e_fwd_bwd, fwd, bwd, outs, gouts, gins
    STATS['captured'] += 1
    return e


def _capture_once(G, impl, e, kw, leaves, need_grad, s_uni):
    fwd = torch.cuda.CUDAGraph()
    with _aliased_leaves(G, leaves) as aliases:
        with H.capture_guard(), torch.cuda.graph(fwd, capture_error_mode='thread_local'), H.zero_arena(e.arena_f):
            with torch.set_grad_enabled(need_grad):
                out = impl(e.s_ws, e.s_c, render_uniforms=s_uni, **kw)
    img, raw = out['image'], out['image_raw']
    img4, raw4 = getattr(img, '_eg3d_padded4', None), getattr(raw, '_eg3d_padded4', None)
    outs = [img4 if img4 is not None else img, raw4 if raw4 is not None else raw, out['image_depth']]
    req = [bool(need_grad and o.requires_grad) for o in outs]
    bwd = gins = gouts = None
    targets = ([e.s_ws] if e.ws_req else []) + ([e.s_c] if e.c_req else []) + list(aliases)
    if any(req):
        gouts = [torch.zeros_like(o) for o, r in zip(outs, req) if r]
        bwd = torch.cuda.CUDAGraph()
        with H.capture_guard(), torch.cuda.graph(bwd, pool=fwd.pool(), capture_error_mode='thread_local'), H.zero_arena(e.arena_b):
            gins = torch.autograd.grad([o for o, r in zip(outs, req) if r], targets, gouts, allow_unused=True)
    del aliases, targets
    return fwd, bwd, outs, req, gouts, gins, (img4 is not None, raw4 is not None)


def _wrap(e, outs):
    img, raw, depth = outs
    if e.padded[0]:
        i4 = img
        img = i4[:, :3]
        img._eg3d_padded4 = i4
    if e.padded[1]:
        r4 = raw
        raw = r4[:, :3]
        raw._eg3d_padded4 = r4
    return {'image': img, 'image_raw': raw, 'image_depth': depth}


def synthesis(G, impl, ws, c, render_uniforms=None, **kw):
    """`impl(ws, c, render_uniforms=..., **kw)` is the per-launch synthesis; see the module docstring."""
    if not (ENABLED and getattr(G, 'graph_eager', True)) or not (torch.is_tensor(ws) and ws.is_cuda and c.is_cuda) or H.PROFILER is not None \
            or torch.cuda.is_current_stream_capturing():
        return impl(ws, c, render_uniforms=render_uniforms, **kw)
    kk = _kw_key(kw)
    if kk is None or kw.get('cache_backbone') or kw.get('use_cached_backbone') or kw.get('update_emas') or kw.get('noise_inject') is not None:
        STATS['eager'] += 1
        return impl(ws, c, render_uniforms=render_uniforms, **kw)
    rkey = _render_key(G)
    if rkey is None:
        STATS['eager'] += 1
        return impl(ws, c, render_uniforms=render_uniforms, **kw)
    leaves, frozen_sig, trainable = _leaves(G)
    grad_on = torch.is_grad_enabled()
    need_grad = bool(grad_on and (ws.requires_grad or c.requires_grad or leaves))
    if not need_grad:
        leaves = []
    pack_inside = bool(trainable)
    uni_key = None if render_uniforms is None else tuple(None if u is None else tuple(u.shape) for u in render_uniforms)
    key = (tuple(ws.shape), tuple(c.shape), ws.dtype, c.dtype, need_grad, bool(ws.requires_grad and need_grad), bool(c.requires_grad and need_grad), kk, uni_key,
           tuple(id(t) for t in leaves), frozen_sig, trainable, H.CONV_MODE, H.MODCONV_OVERRIDE, G.neural_rendering_resolution, ws.device.index, G.training, rkey)
    st = _STATE.get(G)
    if st is None:
        st = _STATE[G] = _PerG()
    e = st.entries.get(key)
    if e is None:
        n = st.counts.get(key, 0) + 1
        st.counts[key] = n
        if n <= HOT_AFTER or key in st.failed:
            if len(st.counts) > 64:
                st.counts.clear()
            STATS['eager'] += 1
            return impl(ws, c, render_uniforms=render_uniforms, **kw)
        try:
            e = _capture(G, impl, ws, c, render_uniforms, kw, leaves, need_grad, pack_inside)
        except RuntimeError as err:          # the runtime refused the capture: keep launching kernel by kernel (same kernels), say so once
            import warnings
            st.failed.add(key)
            STATS['capture_failed'] += 1
            warnings.warn(f'G.synthesis: HIP graph capture failed ({err}); continuing with per-kernel launches')
            torch.cuda.synchronize()
            return impl(ws, c, render_uniforms=render_uniforms, **kw)
        while len(st.entries) >= MAX_ENTRIES:
            st.entries.pop(next(iter(st.entries)))
        st.entries[key] = e
    if not e.done and e.pending is not None and e.pending() is not None:
        # the previous forward of this signature still waits for its backward: its activations live in the static buffers
        STATS['fallback_pending'] += 1
        return impl(ws, c, render_uniforms=render_uniforms, **kw)
    STATS['replayed'] += 1
    u1, u2 = (render_uniforms if render_uniforms is not None else (None, None))
    if need_grad:
        outs = _GraphedFn.apply(e, ws, c, u1, u2, *leaves)
    else:
        with torch.no_grad():
            e.s_ws.copy_(ws)
            e.s_c.copy_(c)
            if e.s_u1 is not None:
                e.s_u1.copy_(u1)
                if e.s_u2 is not None:
                    e.s_u2.copy_(u2)
            e.fwd.replay()
            outs = tuple(o.clone(memory_format=torch.preserve_format) for o in e.s_out)
    return _wrap(e, outs)


def reset(G=None):
    """Drop captured graphs (of one generator, or all)."""
    if G is None:
        _STATE.clear()
    else:
        _STATE.pop(G, None)
