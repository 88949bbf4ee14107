"""Autograd-level fused operators of the MI355X path (each forward/backward = a handful of HIP launches).

These are what the generator modules call instead of the reference's chains of ATen ops:
  ModConvLayerFn  <- SynthesisLayer.forward  (training/networks_stylegan2.py:311-330): style-modulated 3x3 conv (up 1|2) +
                     demod + noise + bias + lrelu + gain + clamp, backward into x, styles, noise_const / strength, bias, weight.
  ToRGBFn         <- ToRGBLayer.forward (:353-357) fused with the skip-image accumulation of SynthesisBlock (:455-457).
  UpsampleImgFn   <- upfirdn2d.upsample2d on the skip image (:453).
  RayGenFn / RenderFn <- RaySampler.forward / ImportanceRenderer.forward.
Formulation: activations are scaled by the styles inside the conv's operand load and the demodulation coefficient is
applied in the epilogue (the reference's non-fused branch, :70-79), so weights are shared by the whole batch and the
style gradient is a reduction fused into the data-gradient conv (no per-sample weight gradient).
"""
import math
import weakref
from typing import Optional

import os
import torch

from . import _lib as L
from . import hipops as H

_F44 = {}


_MINMAX = {}


def _minmax_init(device):
    key = str(device)
    if key not in _MINMAX:
        _MINMAX[key] = torch.tensor([float('inf'), float('-inf')], device=device)
    return _MINMAX[key]


def fir44(device):
    """outer([1,3,3,1])/64 on `device` (upfirdn2d.setup_filter([1,3,3,1]))."""
    key = str(device)
    if key not in _F44:
        f = torch.tensor([1., 3., 3., 1.], dtype=torch.float32, device=device)
        f = torch.outer(f, f)
        _F44[key] = (f / f.sum()).contiguous()
    return _F44[key]


class WeightCache:
    """Packed / derived weight images, rebuilt only when the parameter changes (key: data_ptr + in-place version)."""

    def __init__(self):
        self._c = {}
        self._pad, self._pad_key, self._pad_bias_key = None, None, None

    @staticmethod
    def key_of(w: torch.Tensor):
        # the optimiser epoch only concerns weights an optimiser can write (as hipops.memo): the images of frozen weights survive other
        # generators' steps and captures -- graphs that baked their addresses stay valid
        return (w.data_ptr(), w._version, tuple(w.shape), H.WEIGHTS_EPOCH if w.requires_grad else -1)

    def get(self, w: torch.Tensor):
        key = self.key_of(w)
        hit = self._c.get('k')
        if hit != key:
            with torch.no_grad():
                wf, wa, wsq = H.pack_conv_weight(w)
            self._c = {'k': key, 'wf': wf, 'wa': wa, 'wsq': wsq}
        H.keep_for_capture(self._c['wf'], self._c['wa'], self._c['wsq'])
        return self._c['wf'], self._c['wa'], self._c['wsq']

    def get_split(self, w: torch.Tensor):
        """(forward, adjoint) split images of the packed weights for the pre-split conv kernel (csrc/conv_v2.hip), rebuilt with them."""
        wf, wa, _ = self.get(w)
        hit = self._c.get('split')
        if hit is None:
            o, i, kh, kw = w.shape
            with torch.no_grad():
                hit = (H.split_weight(wf, o, i, kh * kw), H.split_weight(wa, i, o, kh * kw))
            self._c['split'] = hit
        H.keep_for_capture(hit[0].data, hit[0].scale, hit[1].data, hit[1].scale)
        return hit

    def get_pieces(self, w: torch.Tensor):
        """(forward, adjoint) pre-split images of the packed weights for the loader-split conv kernel (hipops.split_weight_pieces): the
        weight-side half of its split arithmetic done once per weight instead of once per workgroup and K-step.  None when a packed
        matrix is not a multiple of four floats."""
        wf, wa, _ = self.get(w)
        hit = self._c.get('pieces')
        if hit is None:
            with torch.no_grad():
                ok = wf.shape[1] % 4 == 0 and wa.shape[1] % 4 == 0
                hit = (H.split_weight_pieces(wf), H.split_weight_pieces(wa)) if ok else (None, None)
            self._c['pieces'] = hit
        H.keep_for_capture(*hit)
        return hit

    def get_padded(self, w: torch.Tensor, cp: int, bias: Optional[torch.Tensor] = None):
        """(wf_p [cp, taps*Ci], wa_p [Ci, taps*cp], bias_p [cp] | None): the packed images with the output-channel dimension zero-padded to
        cp (toRGB 3 -> 4: the padded output channel is 0 + skip, the launch takes the 16-byte vector epilogue, and the data gradient
        contracts over 4 channels).  The buffers are zero-filled once and re-packed in place whenever the weights change -- one launch per
        step of the pivotal-tuning phase instead of a fill and a copy per image."""
        key = self.key_of(w) + (cp,)
        pad = self._pad
        if pad is None or pad[0].shape[0] != cp or pad[0].device != w.device:
            o, i, kh, kw = w.shape
            pad = self._pad = (torch.zeros((cp, kh * kw * i), device=w.device), torch.zeros((i, kh * kw * cp), device=w.device),
                               torch.zeros((cp,), device=w.device))
            self._pad_key = self._pad_bias_key = None
        if self._pad_key != key:
            with torch.no_grad():
                H.pack_conv_weight_padded(w, pad[0], pad[1], cp)
            self._pad_key = key
        if bias is not None:          # (the backward asks without the bias: its image keeps its own key)
            bkey = (bias.data_ptr(), bias._version, H.WEIGHTS_EPOCH if bias.requires_grad else -1)
            if self._pad_bias_key != bkey:
                with torch.no_grad():
                    pad[2][:bias.shape[0]].copy_(bias.detach().float())
                self._pad_bias_key = bkey
        H.keep_for_capture(*pad)
        return pad[0], pad[1], (pad[2] if bias is not None else None)


def prepack_weights(layers):
    """Re-pack the derived weight images of every layer whose weights changed since its last pack in ONE launch (pivotal tuning: all
    generator weights change together, once per step -- 28 launches of a few microseconds otherwise).  `layers`: modules with `.weight`
    [O,I,kh,kw] and `._cache` (WeightCache); an `out_pad` attribute (toRGB: 3 -> 4 channels) selects the zero-padded images.  Layers whose
    images are current (frozen weights) cost one key comparison; the per-layer lazy paths stay valid for anything not passed here."""
    items, installs = [], []
    for m in layers:
        w, cache = m.weight, m._cache
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            continue
        o, i, kh, kw = w.shape
        cp = (o + 3) // 4 * 4
        key = WeightCache.key_of(w)
        if getattr(m, 'out_pad', False) and cp != o:
            pad = cache._pad
            if pad is None or pad[0].shape[0] != cp or pad[0].device != w.device:
                pad = cache._pad = (torch.zeros((cp, kh * kw * i), device=w.device), torch.zeros((i, kh * kw * cp), device=w.device),
                                    torch.zeros((cp,), device=w.device))
                cache._pad_key = cache._pad_bias_key = None
            if cache._pad_key != key + (cp,):
                items.append((w.detach(), pad[0], pad[1], None, cp))
                installs.append((cache, 'pad', key + (cp,), None, False, None))
        elif cache._c.get('k') != key:
            wf = torch.empty((o, kh * kw * i), device=w.device)
            wa = torch.empty((i, kh * kw * o), device=w.device)
            wsq = torch.empty((o, i), device=w.device)
            items.append((w.detach(), wf, wa, wsq, 0))
            installs.append((cache, 'full', key, (wf, wa, wsq), 'split' in cache._c, (o, i, kh * kw)))
    if not items:
        return 0
    H.pack_conv_weights_batched(items)
    resplit = []
    for cache, kind, key, out, had_split, oit in installs:
        if kind == 'pad':
            cache._pad_key = key
        else:
            cache._c = {'k': key, 'wf': out[0], 'wa': out[1], 'wsq': out[2]}
            if had_split:           # the layer ran on the pre-split kernel with its previous weights: its two operand images, batched with the others'
                resplit.append((cache, out, oit))
    if resplit:
        mats = []
        for cache, (wf, wa, _), (o, i, t) in resplit:
            mats += [(wf, o, i, t), (wa, i, o, t)]
        imgs = H.split_weights_batched(mats)
        for k, (cache, _, _) in enumerate(resplit):
            cache._c['split'] = (imgs[2 * k], imgs[2 * k + 1])
    return len(items)


def _zeros_views(device, *shapes):
    """Several small zero-initialised accumulators (targets of atomics) from ONE allocation and ONE fill launch.  A shape of
    None yields None.  Every view starts on a 16-byte boundary."""
    sizes = [0 if s is None else -(-int(torch.Size(s).numel()) // 4) * 4 for s in shapes]
    flat = H.zeros((max(sum(sizes), 1),), device)
    out, o = [], 0
    for s, n in zip(shapes, sizes):
        out.append(None if s is None else flat[o:o + torch.Size(s).numel()].view(s))
        o += n
    return out


TORGB4_ELEMENTWISE = os.environ.get('EG3D_TORGB4_ELEMENTWISE', '1') != '0'   # 4-output toRGB data gradient as an element-wise pass (hipops.torgb_dgrad_act)
USE_PIECES = os.environ.get('EG3D_WEIGHT_PIECES', '1') != '0'      # loader-split conv kernel reads pre-split weight images (WeightCache.get_pieces)
GRAM_FUSED = os.environ.get('EG3D_GRAM_FUSED', '1') != '0'        # decoder-weight gradients inside the renderer's backward kernel (0: operand dumps + hipops.rows_gram)
RENDER_PIPELINE_NOGRAD = os.environ.get('EG3D_RENDER_PIPELINE_NOGRAD', '1') != '0'   # ... also for no-grad rendering (scratch rows)
RENDER_FEAT_ROWS = os.environ.get('EG3D_RENDER_FEAT_ROWS', '1') != '0'   # gather pass + feature rows (eg3d_render_params.feat_rows) in the pipelined renderer
RENDER_PIPELINE = os.environ.get('EG3D_RENDER_PIPELINE', '1') != '0'     # forward renderer as positions -> MFMA decode -> importance -> decode -> composite
KS_TARGET = 256       # blocks a split launch aims for (one per CU; 512 measured 0.7 % slower per step)


def _auto_ksplit(classes, N, Nc, Ck):
    """Split-K factor of an implicit GEMM whose output grid is too small to keep 256 CUs busy (the 4^2..64^2 layers): with few
    128x128 tiles each workgroup walks a long K = taps x channels chain on its own and the launch is latency-bound (64^2 x 512
    channels: 91 TF unsplit, 148 TF split 4 ways).  Slices accumulate with fp32 atomics into a zeroed buffer."""
    blocks = sum((N * c.Ha * c.Wa + 127) // 128 for c in classes) * ((Nc + 127) // 128)
    if blocks >= 200:           # measured: splitting layers with 256 tiles (128^2 x 256 ch) costs more in zero-fill + finish passes than it gains
        return 1
    steps = ((Ck + 15) // 16) * min(c.ntaps for c in classes)
    # 64^2 x 512 (128 tiles, one tap class): 4 slices beat 2 (198 vs 164 TFLOP/s stand-alone, +0.2 % per step); smaller grids keep the target
    target = KS_TARGET * 2 if (len(classes) == 1 and blocks >= 64) else KS_TARGET
    return max(1, min(-(-target // blocks), steps // 8))


# ---- cross-layer backward fusion (EG3D_EPI_BWD_ACT) ------------------------------------------------------------------------------------
# A modulated conv layer's backward starts with an element-wise pass over (dout, out) -> dz + four reductions (eg3d_modconv_epilogue_bwd:
# 3 x tensor bytes, 0.64 ms of a C2 step).  dout is itself the output of the CONSUMER's data-gradient conv, whose epilogue already reads
# this layer's saved output (its `xin`, for the style gradient).  So a layer whose output has exactly one consumer leaves a record at
# forward time; the consumer's backward finds it through its input tensor, launches its data gradient with EPI_BWD_ACT -- which applies
# this layer's activation backward in the same epilogue -- and hands back dz in place of dout.  This layer's backward recognises the
# buffer and skips its own pass.
FUSE_ACT_BWD = os.environ.get('EG3D_FUSE_ACT_BWD', '1') != '0'
FIR_ADJ_LDS = 4096       # pixels from which the FIR adjoint of an up layer's backward runs on the LDS-tiled separable kernel (0 = never)
FUSE_SKIP_UP = os.environ.get('EG3D_FUSE_SKIP_UP', '1') != '0'   # skip image up-sampled inside the toRGB conv's epilogue (eg3d_conv_params::addend_up2)
FUSE_SKIP_ADD = True          # clamped toRGB layers (the SR head): img = upsample2d(img) + y in one pass (eg3d_upfirdn2d_nhwc_add) instead of an up-sampling pass and an add
SPLIT_DZ = os.environ.get('EG3D_SPLIT_DZ', '1') != '0'         # ... and write dz as the data gradient's fp16 operand image where it can (torgb_dgrad_act_split)
_DX_AMAX = {}                 # dx.data_ptr() -> (device scalar max|dx| reported by the data-gradient kernel that wrote it, weak ref to dx); read once by a toRGB backward
_DZ_TOKEN = {}                # device -> 1-element tensor: expanded, it stands in for a dz that only exists as an operand image


def _dz_token(shape, dev):
    t = _DZ_TOKEN.get(dev)
    if t is None:
        t = _DZ_TOKEN[dev] = torch.zeros(1, device=dev)
    return t.expand(shape)


class _ActProducer:
    """`out_ptr` / `shape` identify the layer's output (the tensor itself is NOT held: it is an output of the autograd node that owns this
    record, and a node -> record -> output cycle breaks graph teardown); `node` is a weak reference to that node -- while it is alive its
    saved output is, so the address cannot have been recycled."""
    __slots__ = ('out_ptr', 'shape', 'node', 'd', 'nz', 'nstride', 'noise_strength', 'b', 'gain', 'clamp', 'need', 'fused', 'split_ok')


_PRODUCER_BY_LAYER = {}       # id(layer cache) -> record (at most one per layer: replaced by the layer's next forward)
_PRODUCER_BY_PTR = {}         # out.data_ptr() -> record (the record holds `out`, so the address cannot be recycled while it is listed)


def _set_producer(cache, rec):
    old = _PRODUCER_BY_LAYER.pop(id(cache), None)
    if old is not None and _PRODUCER_BY_PTR.get(old.out_ptr) is old:
        del _PRODUCER_BY_PTR[old.out_ptr]
    if rec is not None:
        _PRODUCER_BY_LAYER[id(cache)] = rec
        _PRODUCER_BY_PTR[rec.out_ptr] = rec
    return old


def _act_bwd_for(x, dev):
    """(record, ActBwdSpec, accumulators) when `x` is the output of a layer that left a producer record, else (None, None, None)."""
    rec = _PRODUCER_BY_PTR.get(x.data_ptr()) if FUSE_ACT_BWD else None
    if rec is None or rec.fused is not None or rec.shape != tuple(x.shape) or rec.node() is None:
        return None, None, None
    N, Co = x.shape[:2]
    need_b, need_dd, need_nz, need_ns = rec.need
    has_nz = rec.nz is not None
    acc = _zeros_views(dev, (Co,) if need_b else None, (N, Co) if need_dd else None, tuple(rec.nz.shape) if (need_nz and has_nz) else None,
                       () if (need_ns and has_nz) else None, (1,))
    spec = H.ActBwdSpec(d=rec.d, bias=rec.b, noise=rec.nz, noise_nstride=rec.nstride or 0, noise_strength=rec.noise_strength if has_nz else None,
                        act='lrelu', alpha=0.2, gain=rec.gain, clamp=rec.clamp, dbias=acc[0], dd=acc[1], dnoise=acc[2],
                        dnoise_nstride=rec.nstride or 0, dstrength=acc[3])
    return rec, spec, acc


# Unfinished split-K data gradients on their way to a toRGB node (hipops.DEFER_DGRAD_FINISH): key = data_ptr of the tensor both nodes know (the
# conv's input x = the toRGB node's x), value = (z, styles, ds).  Written by ModConvLayerFn.backward, consumed by the ToRGBFn.backward that runs next in
# the same backward pass; that node finishes the gradient inside its own launch -- or with eg3d_dgrad_finish when it cannot.
def _igemm_precision():
    """Arithmetic of the launches that stay on the loader-split kernel (toRGB): it has no single-product form, so under
    hipops.modconv_override('f16x1') they keep three products."""
    pr = H.modconv_precision()
    return 'f16x3' if pr == 'f16x1' else pr


PENDING_DGRAD = {}


def _pend_dgrad(x, z, styles, ds):
    """Queue an unfinished split-K data gradient for the toRGB node that receives it next IN THIS backward pass.  Entries are keyed by the
    layer input's address, so they must not outlive the pass: the first entry of a pass registers an engine callback that clears the table
    when the pass completes (ToRGBFn.forward clears it as well -- an abandoned pass, where callbacks do not run, is covered by that)."""
    if not PENDING_DGRAD:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(PENDING_DGRAD.clear)
        except Exception:          # not inside an engine-driven backward (a direct call in a test): the forward-time clear remains
            pass
    PENDING_DGRAD[x.data_ptr()] = (z, styles, ds)


def _finish_pending(pend, x):
    z, s0, ds0 = pend
    fin = H.empty_cl(*x.shape, x.device)
    H.dgrad_finish(z, x, s0, fin, ds=ds0)
    return fin


class ModConvLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, styles, noise, noise_strength, bias, up, act_gain, clamp, cache, want_wgrad, d_in=None, single_consumer=False,
                input_is_layer_output=False, precision=None, next_styles=None, defer_epilogue=False, rgb_head=None):
        # rgb_head: (toRGB weight [O<=4, Co, 1, 1], its styles [N, Co], bias | None, clamp | None, its WeightCache) of the toRGB layer that reads this
        # layer's output next: evaluated in this launch's epilogue where the kernel can (hipops.conv_v2 rgb_head); the result rides on the output
        # tensor (`_eg3d_rgb_y`) for ToRGBFn.forward, which then launches nothing
        # defer_epilogue: the caller hands the output to a toRGB node NEXT and to nothing before it: a split-K layer then leaves its finishing pass
        # to that node's launch (hipops.PendingEpilogue on the output tensor; ToRGBFn.forward runs or absorbs it)
        # precision: None = the process-wide arithmetic of the modulated convs; 'f16x1' = the reference's fp16 layers (one product of
        # fp16-rounded operands, fp32 accumulation; forward, data gradient and weight gradient alike)
        # x: CL [N,Ci,H,W]; weight [Co,Ci,3,3]; styles [N,Ci]; noise None | [res,res] | [N,1,res,res]; noise_strength 0-d
        # d_in: the demodulation coefficients [N,Co] when the style bank already computed them (their gradient is then returned)
        L.require_cuda(x, weight, styles)
        from_torgb = bool(getattr(x, '_eg3d_from_torgb', False)) and H.is_cl(x) and x.dtype == torch.float32      # x is a toRGB node's pass-through output (SynthesisBlock.forward)
        x = H.to_cl(x.float())
        styles = styles.contiguous().float()
        N, Ci, Hi, Wi = x.shape
        Co, _, kh, kw = weight.shape
        wf, wa, wsq = cache.get(weight)
        d = d_in.contiguous().float() if d_in is not None else H.demod_fwd(styles, wsq)
        Ho, Wo = Hi * up, Wi * up
        nz = nstride = None
        if noise is not None:
            nz = noise.contiguous().float()
            nstride = 0 if nz.dim() == 2 else Ho * Wo
        clampv = -1.0 if clamp is None else float(clamp)
        out = H.empty_cl(N, Co, Ho, Wo, x.device)
        b = bias.contiguous().float() if bias is not None else None
        aflops = 2.0 * N * Hi * Wi * (1 if up == 2 else 1) * kh * kw * Ci * Co     # MACs of the (transposed) conv
        pending = None
        prec = precision or H.modconv_precision()
        ig_prec = 'f16x3' if prec == 'f16x1' else prec      # the loader-split kernel has no single-product form: it keeps the three products
        # pre-split weight image for the loader-split kernel; frozen weights only (a trained weight would need the pass every step: +34 launches for ~1 %)
        wfp = cache.get_pieces(weight)[0] if (ig_prec == 'f16x3' and USE_PIECES and not weight.requires_grad) else None
        amax_out = H.zeros((1,), x.device)            # max|out|, reported by whichever kernel writes `out`: the next layer's operand range
        cls, Hz, Wz = (H.classes_corr(Ho, Wo, kh, kw, kh // 2), Ho, Wo) if up == 1 else H.classes_convT(Hi, Wi, kh, kw, up)
        ks = _auto_ksplit(cls, N, Co, Ci)
        # (transposed-conv classes run on it too, but measured slower than the loader-split kernel: three launches of 4 / 2 / 1-tap
        #  classes on ragged 257-wide grids -- 196 vs 174 us on 256^2 x 256 -> 513^2 x 128, 116 vs 67 us on 128^2 x 256; opt-in)
        # (a grid the pre-split kernel fills with its 2-row patches -- 64^2 x 512 -- goes there un-split rather than to a split-K launch)
        v2 = H.USE_V2 and prec in ('f16x3', 'f16x1') and (up == 1 or (H.V2_CONVT and prec == 'f16x3')) and H.conv_v2_supported(Ci, Co, cls, N)
        v2 = v2 and (ks == 1 or (up == 1 and H.conv_v2_rows(Ci, Co, cls, N) == 2))
        if v2:
            ks = 1
        nprod = 1 if prec == 'f16x1' else 3
        # 3x3 layers whose grids cannot fill the chip with 256 x 128 tiles (64^2 x 512, 32^2 x 512 at one image): the wave-split kernel
        # (csrc/conv_v3.hip) -- 128 / 64-cell x 64-channel tiles, the contraction split over the waves of a workgroup, fused epilogue, no zero
        # fill / atomics / finishing pass
        v3p = H.conv_v3_plan(Ci, Co, cls, N) if (up == 1 and prec in ('f16x3', 'f16x1')) else None
        if v3p:
            v2, ks = False, 1
        # ... or (opt-in) the pre-split kernel with the contraction split over workgroups (atomic partial tiles) + the finishing epilogue pass
        ks2 = H.conv_v2_ksplit(Ci, Co, cls, N) if (up == 1 and not v2 and not v3p and prec in ('f16x3', 'f16x1')) else 0
        # up-sampling layers: the four output parities of the transposed conv from one workgroup per input patch (csrc/conv_v2_up.hip)
        ksu = H.conv_up2_plan(Ci, Co, Hi, Wi, N) if (up == 2 and not v2 and kh == 3 and kw == 3 and prec in ('f16x3', 'f16x1')) else None
        epi_kw = dict(noise=nz, noise_nstride=nstride or 0, noise_strength=noise_strength, bias=b, act='lrelu', alpha=0.2, gain=act_gain, clamp=clampv)
        if v2 or v3p or ks2 or ksu:   # pre-split operands: modulation, range normalisation and the fp16 split happen once, not per tile and tap
            pre_img = getattr(x, '_eg3d_split', None)           # (SplitImage, styles ptr, styles version) left by the producing layer's epilogue
            if pre_img is not None and pre_img[1] == styles.data_ptr() and pre_img[2] == styles._version and pre_img[0].shape == tuple(x.shape):
                aimg = pre_img[0]
            else:
                aimg = H.split_activation(x, H.amax_of(x), in_scale=styles)
            wimg = cache.get_split(weight)[0]
        rgb_y = None
        if up == 1:
            if v2:
                rkw = {}
                if rgb_head is not None and H.RGB_HEAD and Co == 128 and len(cls) == 1 and H.conv_v2_rows(Ci, Co, cls, N) == 8:
                    tw, ts, tb, tclamp, tcache = rgb_head
                    if tw.shape[0] <= 4 and tw.shape[1] == Co and tuple(ts.shape) == (N, Co):
                        tw4, _, tb4 = tcache.get_padded(tw, 4, tb) if tw.shape[0] != 4 else (tcache.get(tw)[0], None, tb.contiguous().float() if tb is not None else None)
                        y4 = H.empty_cl(N, 4, Ho, Wo, x.device)
                        ts_c = ts.contiguous().float()
                        rkw = dict(rgb_head=(tw4, ts_c, tb4, y4, -1.0 if tclamp is None else float(tclamp), 3 if tw.shape[0] == 3 else 4))
                        rgb_y = (y4, tw.data_ptr(), tw._version, ts.data_ptr(), ts._version, None if tb is None else (tb.data_ptr(), tb._version),
                                 -1.0 if tclamp is None else float(tclamp))
                ran = []
                H.conv_v2(aimg, wimg, out, cls, epi=L.EPI_FWD, out_scale=d, out_amax=amax_out, algo_flops=aflops, products=nprod, rgb_head_ran=ran, **epi_kw, **rkw)
                if rgb_y is not None and ran != [True]:
                    rgb_y = None                # the library refused the head's operands: ToRGBFn launches the 1x1 layer
            elif v3p:
                H.conv_v3(aimg, wimg, out, cls, plan=v3p, epi=L.EPI_FWD, out_scale=d, out_amax=amax_out, algo_flops=aflops, products=nprod, **epi_kw)
            elif ks2:
                z = H.zeros_cl(N, Co, Ho, Wo, x.device)
                H.conv_v2(aimg, wimg, z, cls, epi=L.EPI_ATOMIC, ksplit=ks2, algo_flops=aflops, products=nprod)
                if defer_epilogue and H.DEFER_EPILOGUE:
                    pending = H.PendingEpilogue(z, out, d, amax_out, **epi_kw)
                else:
                    H.epilogue_fwd(z, out, d=d, out_amax=amax_out, **epi_kw)
            elif ks == 1:
                H.conv_igemm(x, wf, Ci, Co, out, cls, in_scale=styles, epi=L.EPI_FWD, out_scale=d, algo_flops=aflops, precision=ig_prec, out_amax=amax_out, w_pieces=wfp,
                             **epi_kw)
            else:
                z = H.zeros_cl(N, Co, Ho, Wo, x.device)
                if prec in ('f16x3', 'f16x1') and (H.CONV_WS_TRAINABLE or not weight.requires_grad) and H.conv_ws_ok(Ci, Co, cls, N, Hi, Wi):
                    # 4^2 .. 16^2: one workgroup per (channel tile, 16-channel chunk), every weight byte fetched once, operand split inside (csrc/conv_ws.hip)
                    H.conv_ws(x, cache.get_split(weight)[0], z, cls, in_scale=styles, x_amax=H.amax_of(x), products=nprod, algo_flops=aflops)
                else:
                    H.conv_atomic(x, wf, Ci, Co, z, cls, in_scale=styles, ksplit=ks, algo_flops=aflops, precision=ig_prec, w_pieces=wfp)
                if defer_epilogue and H.DEFER_EPILOGUE:
                    pending = H.PendingEpilogue(z, out, d, amax_out, **epi_kw)
                else:
                    H.epilogue_fwd(z, out, d=d, out_amax=amax_out, **epi_kw)
        else:
            if ksu:
                ksplit, ragged, urows = ksu
                z = H.zeros_cl(N, Co, Hz, Wz, x.device) if ksplit > 1 else H.empty_cl(N, Co, Hz, Wz, x.device)
                if ragged:      # the full (Hi + 1) x (Wi + 1) cell grid still fits one round of workgroups
                    H.conv_up2(aimg, wimg, z, epi=L.EPI_ATOMIC if ksplit > 1 else L.EPI_STORE, ksplit=ksplit, products=nprod, algo_flops=aflops, patch_rows=urows)
                else:
                    # main grid (Hi x Wi cells, perfectly tiled) on the fused-parity kernel; the last output row / column (1-D problems, ~20 us
                    # of latency for 0.1 GFLOP) as four small tap classes of the loader-split kernel.  (Forking that launch onto a second
                    # stream hides it at N = 1 -- but two processes sharing one device then replayed the two-branch graph at 1.2 s per step:
                    # not worth the risk on an 8-rank node.)
                    H.conv_up2(aimg, wimg, z, Hc=Hi, Wc=Wi, epi=L.EPI_ATOMIC if ksplit > 1 else L.EPI_STORE, ksplit=ksplit, products=nprod, algo_flops=aflops, patch_rows=urows)
                    H.conv_igemm(x, wf, Ci, Co, z, H.up2_border_classes(Hi, Wi), out_stride=up, in_scale=styles, epi=L.EPI_STORE, algo_flops=0.0,
                                 precision=ig_prec, w_pieces=wfp)
            elif v2:
                z = H.empty_cl(N, Co, Hz, Wz, x.device)
                H.conv_v2(aimg, wimg, z, cls, out_stride=up, epi=L.EPI_STORE, algo_flops=aflops)
            elif ks == 1:
                z = H.empty_cl(N, Co, Hz, Wz, x.device)
                H.conv_igemm(x, wf, Ci, Co, z, cls, out_stride=up, in_scale=styles, epi=L.EPI_STORE, algo_flops=aflops, precision=ig_prec, w_pieces=wfp)
            else:
                z = H.zeros_cl(N, Co, Hz, Wz, x.device)
                if up == 2 and kh == 3 and kw == 3 and prec in ('f16x3', 'f16x1') and (H.CONV_WS_TRAINABLE or not weight.requires_grad) and H.conv_ws_up_ok(Ci, Co, N, Hi, Wi):
                    # 4^2 / 8^2 input cells: the weight-streaming kernel's transposed form (four parity accumulator sets per wave, csrc/conv_ws.hip)
                    H.conv_ws_up(x, cache.get_split(weight)[0], z, in_scale=styles, x_amax=H.amax_of(x), products=nprod, algo_flops=aflops)
                else:
                    H.conv_atomic(x, wf, Ci, Co, z, cls, out_stride=up, in_scale=styles, ksplit=ks, algo_flops=aflops, precision=ig_prec, w_pieces=wfp)
            simg = None
            if H.UPCONV_EPI and Co % 64 == 0 and up == 2:
                # consumer = a 3x3 layer on the pre-split kernel (same channel count, output resolution): its operand image comes out of this
                # epilogue when the range is known beforehand (conv_clamp: the super-resolution head)
                want_split = (H.UPCONV_EPI_SPLIT and next_styles is not None and clampv >= 0 and prec in ('f16x3', 'f16x1') and H.USE_V2
                              and tuple(next_styles.shape) == (N, Co) and H.conv_v2_supported(Co, Co, H.classes_corr(Ho, Wo, 3, 3, 1), N))
                ns = next_styles.contiguous().float() if want_split else None
                simg = H.upconv_epilogue_fwd(z, out, pad0=1, fir_gain=float(up * up), d=d, out_amax=amax_out, split_in_scale=ns, **epi_kw)
                if simg is not None:
                    out._eg3d_split = (simg, ns.data_ptr(), ns._version)
            else:
                H.epilogue_fwd(z, out, fir=fir44(x.device), pad0=1, fir_gain=float(up * up), d=d, out_amax=amax_out, **epi_kw)
        H.tag_amax(out, amax_out)
        if rgb_y is not None:
            out._eg3d_rgb_y = rgb_y + (out._version,)          # (ToRGBFn.forward accepts y only for THIS content of the output and the styles)
        if pending is not None:
            out._eg3d_pending_epi = pending
        rec = None
        if FUSE_ACT_BWD and single_consumer and any(ctx.needs_input_grad[:6]):      # see _ActProducer: the consumer may run this layer's activation backward
            ng = ctx.needs_input_grad
            rec = _ActProducer()
            rec.out_ptr, rec.shape, rec.node = out.data_ptr(), tuple(out.shape), weakref.ref(ctx)
            rec.d, rec.nz, rec.nstride, rec.noise_strength, rec.b, rec.gain, rec.clamp, rec.fused = d, nz, nstride, noise_strength, b, act_gain, clampv, None
            rec.need = (bool(ng[5]), bool(ng[2] or (ng[1] and want_wgrad)), bool(ng[3]), bool(ng[4]))
            # this layer's backward will feed dz to the pre-split data-gradient kernel and to nothing else (frozen weights): a consumer that
            # runs the activation backward may then hand dz over as that kernel's operand image instead of an fp32 tensor (SPLIT_DZ)
            rec.split_ok = False
            if SPLIT_DZ and up == 1 and not (ng[1] and want_wgrad) and prec in ('f16x3', 'f16x1') and H.USE_V2 and Co % 8 == 0 and (ng[0] or ng[2]):
                cls_adj0 = H.classes_corr_adjoint(Hi, Wi, kh, kw, kh // 2)
                rec.split_ok = bool((_auto_ksplit(cls_adj0, N, Ci, Co) == 1 or H.conv_v2_rows(Co, Ci, cls_adj0, N) == 2) and H.conv_v2_supported(Co, Ci, cls_adj0, N))
        if rec is not None:             # (a no-grad forward of the same layer -- the canonical view of the warping loss -- leaves a pending record alone)
            _set_producer(cache, rec)
        # pivotal tuning: the weight gradient reads the forward's operand image again (csrc/conv_wgrad_v2.hip) instead of the fp32 activation
        ctx.aimg = aimg if ((((v2 or v3p) and up == 1) or (ksu and up == 2)) and want_wgrad and ctx.needs_input_grad[1] and H.WGRAD_V2 and Ci % 64 == 0 and Co % 64 == 0) else None
        ctx.rec = rec                   # THIS forward's record: the backward below trusts only it (two live graphs of one layer cannot mix)
        ctx.save_for_backward(x, weight, styles, d, out, nz, noise_strength, b)
        ctx.cfg = (up, act_gain, clampv, nstride, cache, want_wgrad, noise is not None and noise.dim() == 4, d_in is not None)
        ctx.prec = prec
        # both ends opt in: the producer promises a single consumer, the consumer that its x is that producer's output handed over directly
        # (NOT the copy routed through a toRGB node: that gradient is summed inside the toRGB data gradient, which is the fusing launch then)
        ctx.fuse_input = bool(input_is_layer_output)
        ctx.from_torgb = from_torgb
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, styles, d, out, nz, noise_strength, b = ctx.saved_tensors
        up, act_gain, clampv, nstride, cache, want_wgrad, noise4d, d_given = ctx.cfg
        need_x, need_w, need_s, need_nz, need_ns, need_b = ctx.needs_input_grad[:6]
        need_w = need_w and want_wgrad
        rec = ctx.rec
        is_token = (rec is not None and rec.fused is not None and len(rec.fused) > 6 and rec.fused[0].data_ptr() == dout.data_ptr()
                    and rec.fused[0].shape == dout.shape)          # dz exists only as an operand image (see _dz_token): nothing to make contiguous
        if not is_token:
            dout = H.to_cl(dout.float())
        N, Ci, Hi, Wi = x.shape
        Co, _, kh, kw = weight.shape
        Ho, Wo = Hi * up, Wi * up
        dev = x.device
        wf, wa, wsq = cache.get(weight)
        if rec is not None:
            if _PRODUCER_BY_LAYER.get(id(cache)) is rec:
                del _PRODUCER_BY_LAYER[id(cache)]
            if _PRODUCER_BY_PTR.get(rec.out_ptr) is rec:
                del _PRODUCER_BY_PTR[rec.out_ptr]
        pre = None              # (dz, dbias, dd, dnoise, dstrength, amax) when the consumer's data gradient already ran this layer's activation backward
        if rec is not None and rec.fused is not None and rec.fused[0].data_ptr() == dout.data_ptr() and rec.fused[0].shape == dout.shape:
            pre = rec.fused
        if rec is not None:
            rec.fused = None    # the accumulators become parameter gradients: AccumulateGrad adopts a gradient nobody else references, copies it otherwise
        dz = pre[0] if pre is not None else H.empty_cl(N, Co, Ho, Wo, dev)
        ks_adj = rep = None
        if need_x or need_s:
            cls_probe = H.classes_corr_adjoint(Hi, Wi, kh, kw, kh // 2) if up == 1 else H.classes_convT_adjoint(Hi, Wi, kh, kw, up)
            ks_adj = _auto_ksplit(cls_probe, N, Ci, Co)
            # thousands of tiles reduce into the same N*Ci style-gradient addresses: spread them over replicas, sum afterwards
            rep = 1          # replicas of the style-gradient accumulator (eg3d_conv_params::ds_replicas) measured no gain on MI355X
        # all small atomically-accumulated outputs of this backward from one zero fill
        prec = ctx.prec
        ig_prec = 'f16x3' if prec == 'f16x1' else prec
        wap = cache.get_pieces(weight)[1] if (ig_prec == 'f16x3' and USE_PIECES and not weight.requires_grad) else None
        dz_img = None
        if pre is not None:
            _, dbias, dd, dnoise, dstrength, amax = pre[:6]
            dz_img = pre[6] if len(pre) > 6 else None
            ds = None if ks_adj is None else H.zeros((N, Ci), dev)
        else:
            dbias, dd, dnoise, dstrength, ds, amax = _zeros_views(
                dev, (Co,) if need_b else None, (N, Co) if (need_s or need_w) else None,
                tuple(nz.shape) if (need_nz and nz is not None) else None, () if (need_ns and nz is not None) else None,
                None if ks_adj is None else ((rep, N, Ci) if rep > 1 else (N, Ci)),
                (1,) if (prec in ('f16x3', 'f16x1') and ks_adj is not None) else None)          # max|dz|: operand range of the two-piece fp16 data gradient
            H.epilogue_bwd(dout, out, dz, d=d, noise=nz, noise_nstride=nstride or 0, noise_strength=noise_strength if nz is not None else None,
                           bias=b, act='lrelu', alpha=0.2, gain=act_gain, clamp=clampv, dbias=dbias, dd=dd, dnoise=dnoise,
                           dnoise_nstride=nstride or 0, dstrength=dstrength, dz_amax=amax)
        amul = 1.0 if up == 1 else float(up * up)       # g = FIR(dz) * up^2 with a non-negative unit-sum filter: |g| <= up^2 max|dz|
        gimg = None
        if up == 1:
            g = dz
            cls_adj = H.classes_corr_adjoint(Hi, Wi, kh, kw, kh // 2)
            in_stride = 1
            cls_w, out_stride_w = H.classes_corr(Ho, Wo, kh, kw, kh // 2), 1
        elif ((need_x or need_s or need_w) and up == 2 and kh == 3 and kw == 3 and prec in ('f16x3', 'f16x1') and amax is not None
              and (not (need_x or need_s) or H.conv_s2adj_ok(Co, Ci, Hi, Wi, N) or H.conv_v3_s2adj_ok(Co, Ci, Hi, Wi, N))
              and (not need_w or H.conv_wgrad_v2_up_ok(Ci, Co, Hi, Wi, N))):
            # the FIR adjoint writes the gradient operand directly as parity-split fp16 images (range bound up^2 max|dz|) -- no fp32 g, no strided
            # gathers in a conv loader: the data gradient (conv_v2_s2adj / conv_v3_s2adj) and, when the weights train, the weight gradient
            # (conv_wgrad_v2_up) both read them
            g = None
            gimg = H.fir44_adjoint_split(dz, amax, gain=float(up * up))
            cls_adj = H.classes_convT_adjoint(Hi, Wi, kh, kw, up)
            in_stride = up
            cls_w, out_stride_w = H.classes_convT(Hi, Wi, kh, kw, up)[0], up
        else:
            if FIR_ADJ_LDS and Co % 64 == 0 and dz.shape[2] * dz.shape[3] >= FIR_ADJ_LDS:
                # the separable, LDS-tiled FIR pass of the forward (1.6 loads per output instead of 6.25): the [1,3,3,1] filter is its own flip
                g = H.empty_cl(N, Co, dz.shape[2] + 1, dz.shape[3] + 1, dev)
                H.upconv_epilogue_fwd(dz, g, pad0=2, fir_gain=float(up * up))
            else:
                g = H.upfirdn2d_nhwc(dz, fir44(dev), pad=(2, 2, 2, 2), flip=True, gain=float(up * up))     # adjoint of the FIR
            cls_adj = H.classes_convT_adjoint(Hi, Wi, kh, kw, up)
            in_stride = up
            cls_w, out_stride_w = H.classes_convT(Hi, Wi, kh, kw, up)[0], up
        dx = None
        if need_x or need_s:
            dx = H.empty_cl(N, Ci, Hi, Wi, dev)
            aflops = 2.0 * N * Hi * Wi * kh * kw * Ci * Co
            ks = ks_adj
            # x is some layer's output: if that layer left a record, this launch also runs ITS activation backward (dx then holds its dz)
            prod, spec, pacc = _act_bwd_for(x, dev) if (need_x and ctx.fuse_input) else (None, None, None)
            fkw = dict(act_bwd=spec, out_amax=pacc[4]) if prod is not None else {}
            ks2 = H.conv_v2_ksplit(Co, Ci, cls_adj, N) if (up == 1 and prec in ('f16x3', 'f16x1') and amax is not None) else 0
            if gimg is not None:
                if not fkw and SPLIT_DZ:          # dx goes on to a toRGB node as its pass-through gradient: that pass wants max|dx| (torgb_dgrad_act_split)
                    dx_amax = H.zeros((1,), dev)
                    fkw = dict(out_amax=dx_amax)
                    _DX_AMAX[dx.data_ptr()] = (dx_amax, weakref.ref(dx))
                did = H.conv_v2_s2adj(gimg, cache.get_split(weight)[1], dx, cls_adj, epi=L.EPI_BWD, out_scale=styles, xin=x, ds=ds, algo_flops=aflops,
                                      products=1 if prec == 'f16x1' else 3, v3=not H.conv_s2adj_ok(Co, Ci, Hi, Wi, N), **fkw)
            elif H.USE_V2 and up == 1 and (ks == 1 or H.conv_v2_rows(Co, Ci, cls_adj, N) == 2) and prec in ('f16x3', 'f16x1') and H.conv_v2_supported(Co, Ci, cls_adj, N):
                gimg = dz_img if dz_img is not None else H.split_activation(g, amax)           # (kept: the weight gradient below reads it too)
                did = H.conv_v2(gimg, cache.get_split(weight)[1], dx, cls_adj, epi=L.EPI_BWD,
                                out_scale=styles, xin=x, ds=ds, algo_flops=aflops, products=1 if prec == 'f16x1' else 3, **fkw)
                dz_img = None
            elif up == 1 and prec in ('f16x3', 'f16x1') and amax is not None and H.conv_v3_plan(Co, Ci, cls_adj, N):
                # under-filled 3x3 grid: data gradient, style gradient and the producer's activation backward from one launch of the wave-split kernel
                gimg = dz_img if dz_img is not None else H.split_activation(g, amax)
                did = H.conv_v3(gimg, cache.get_split(weight)[1], dx, cls_adj, plan=H.conv_v3_plan(Co, Ci, cls_adj, N), epi=L.EPI_BWD, out_scale=styles,
                                xin=x, ds=ds, algo_flops=aflops, products=1 if prec == 'f16x1' else 3, **fkw)
                dz_img = None
            elif ks2:                              # under-filled 3x3 grid: split-K launch of the pre-split kernel, then the finishing pass
                z = H.zeros_cl(N, Ci, Hi, Wi, dev)
                H.conv_v2(H.split_activation(g, amax), cache.get_split(weight)[1], z, cls_adj, epi=L.EPI_ATOMIC, ksplit=ks2, algo_flops=aflops,
                          products=1 if prec == 'f16x1' else 3)
                did = prod is not None and Ci % 4 == 0 and Ci <= 1024
                if did:
                    H.dgrad_finish_act(z, x, styles, dx, spec, ds=ds, dz_amax=pacc[4])
                elif ctx.from_torgb and H.DEFER_DGRAD_FINISH and need_x and prod is None and d_given:
                    _pend_dgrad(x, z, styles, ds)                          # the toRGB node that receives this gradient next finishes it in its launch (d_given: `ds` is read by
                                                                          # the style bank's node, which runs after every layer -- a per-layer affine would read it before the x.z term lands)
                    dx = z
                else:
                    H.dgrad_finish(z, x, styles, dx, ds=ds)
            elif ks == 1:
                did = H.conv_igemm(g, wa, Co, Ci, dx, cls_adj, in_stride=in_stride, epi=L.EPI_BWD, out_scale=styles, xin=x, ds=ds, algo_flops=aflops, w_pieces=wap,
                                   precision=ig_prec, a_amax=amax, a_amax_mul=amul, **fkw)
                if rep > 1:
                    ds = ds.sum(0)
            else:                                  # low resolution: split K over blocks, then scale / reduce in a finishing pass
                z = H.zeros_cl(N, Ci, Hi, Wi, dev)
                if up == 1 and prec in ('f16x3', 'f16x1') and amax is not None and (H.CONV_WS_TRAINABLE or not weight.requires_grad) and H.conv_ws_ok(Co, Ci, cls_adj, N, Hi, Wi):
                    H.conv_ws(g, cache.get_split(weight)[1], z, cls_adj, x_amax=amax, products=1 if prec == 'f16x1' else 3, algo_flops=aflops)
                elif (up == 2 and g is not None and kh == 3 and kw == 3 and prec in ('f16x3', 'f16x1') and amax is not None and (H.CONV_WS_TRAINABLE or not weight.requires_grad)
                      and H.conv_ws_ok(Co, Ci, cls_adj, N, Hi, Wi, in_stride=2)):
                    H.conv_ws(g, cache.get_split(weight)[1], z, cls_adj, x_amax=amax, x_amax_mul=amul, products=1 if prec == 'f16x1' else 3, algo_flops=aflops,
                              in_stride=2)
                else:
                    H.conv_atomic(g, wa, Co, Ci, z, cls_adj, in_stride=in_stride, ksplit=ks, algo_flops=aflops, precision=ig_prec, a_amax=amax, w_pieces=wap,
                                  a_amax_mul=amul)
                did = prod is not None and Ci % 4 == 0 and Ci <= 1024
                if did:
                    H.dgrad_finish_act(z, x, styles, dx, spec, ds=ds, dz_amax=pacc[4])
                elif ctx.from_torgb and H.DEFER_DGRAD_FINISH and need_x and prod is None and d_given:
                    _pend_dgrad(x, z, styles, ds)                          # the toRGB node that receives this gradient next finishes it in its launch
                    dx = z
                else:
                    H.dgrad_finish(z, x, styles, dx, ds=ds)
            if prod is not None and did is True:
                prod.fused = (dx,) + tuple(pacc)
        if dz_img is not None or (is_token and need_w):
            raise RuntimeError('ModConvLayerFn.backward: dz was handed over as an operand image but this backward needs the fp32 tensor '
                               '(set EG3D_SPLIT_DZ=0 and report the configuration)')
        # with d from the style bank, dd is returned and its d styles part handled there; the d weight part of it is in weight_grad_finish below
        if dd is not None and need_s and not d_given:
            if ds is None:
                ds = H.zeros((N, Ci), dev)
            H.demod_bwd(styles, wsq, d, dd, ds=ds)
        dweight = None
        if need_w:
            dwp = None
            # same arithmetic as the data gradient: two-piece fp16 split with the gradient operand range-normalised by max|dz|
            wprec = prec if (prec in ('f16x3', 'f16x1') and amax is not None) else 'f32'
            ximg = getattr(ctx, 'aimg', None)
            if up == 1 and ximg is not None and wprec != 'f32' and amax is not None:
                if gimg is None:
                    gimg = H.split_activation(g, amax)
                use_v2w = H.conv_wgrad_v2_ok(gimg, ximg, cls_w)
            else:
                use_v2w = False
            if up == 2 and g is None:       # up layer on the parity-split path: G as the four parity images, X as the forward's (or a fresh) operand image
                if ximg is None:
                    ximg = H.split_activation(x, H.amax_of(x), in_scale=styles)
                wtaps = [0] * 9
                for c_ in cls_w:
                    for t_ in range(c_.ntaps):
                        wtaps[3 * (c_.out_py - 2 * c_.dy[t_]) + (c_.out_px - 2 * c_.dx[t_])] = c_.wtap[t_]
                dwp = H.conv_wgrad_v2_up(gimg, ximg, H.zeros(wf.shape, dev), wtaps, products=1 if wprec == 'f16x1' else 3)
            elif use_v2w:       # both operands as the split images the forward / data gradient consumed: LDS-DMA + transposing LDS reads, no VALU loader
                if H.WGRAD_SLABS:       # partial tiles stored, summed in slab order by weight_grad_finish: no atomics, no zero fill
                    dwp = H.conv_wgrad_v2_slabs(gimg, ximg, cls_w, products=1 if wprec == 'f16x1' else 3)
                else:
                    dwp = H.conv_wgrad_v2(gimg, ximg, H.zeros(wf.shape, dev), cls_w, products=1 if wprec == 'f16x1' else 3)
            else:
                dwp = H.zeros(wf.shape, dev)
                H.conv_wgrad(x, g, Ci, Co, dwp, cls_w, in_stride=1, out_stride=out_stride_w, in_scale=styles, precision=wprec,
                             g_amax=amax if wprec != 'f32' else None, g_amax_mul=amul)
            # [O,taps,I] accumulator -> the parameter's own (contiguous [O,I,kh,kw]) layout, plus the demodulation path d wsq / d w = 2 w
            # (dwsq from dd on the fly), in one pass; the fused multi-tensor Adam walks parameter and gradient with the same linear index
            dweight = H.weight_grad_finish(dwp, weight, styles, d, dd)
        if dnoise is not None and noise4d:
            dnoise = dnoise.view(N, 1, Ho, Wo)
        return (dx if need_x else None, dweight, ds if need_s else None, dnoise, dstrength, dbias, None, None, None, None, None,
                dd if d_given else None, None, None, None, None, None, None)


class BroadcastRowsFn(torch.autograd.Function):
    """ws = w.repeat([1, L, 1]) of the latent projector (projectors/w_projector.py:117: one optimised row, num_ws identical style inputs) as a
    stride-0 view: no copy forward.  Backward: the style bank, which reads such a view through its single row, delivers the whole gradient in
    row 0 of a zero-filled [N,L,D] tensor and says so (`_eg3d_row0_total`) -- row 0 is returned; any other gradient is summed over the rows."""

    @staticmethod
    def forward(ctx, w, L_):
        ctx.L = int(L_)
        return w.expand(-1, ctx.L, -1)

    @staticmethod
    def backward(ctx, g):
        if getattr(g, '_eg3d_row0_total', False):
            return g[:, :1, :], None
        return g.sum(1, keepdim=True), None


def broadcast_rows(w: torch.Tensor, L_: int) -> torch.Tensor:
    """[N,1,D] -> [N,L,D] (see BroadcastRowsFn)."""
    return BroadcastRowsFn.apply(w, int(L_))


class StyleBankFn(torch.autograd.Function):
    """Styles -- and, for the conv layers, demodulation coefficients -- of all modulated layers of a network from two launches
    (eg3d_style_affine_fwd/_bwd).  apply(ws, plan, *weights_and_biases) -> tuple: the L styles [N, C_l], then one d [N, Co_l] per
    layer whose plan entry carries wsq.  plan = tuple of (wrow, wgain, bgain, post, has_bias, wsq | None) per layer.  Gradients flow
    to ws and to the affine weights / biases that require them (the demodulation part covers d d / d styles; a layer with trainable conv
    weights adds d d / d weight itself)."""

    @staticmethod
    def forward(ctx, ws, plan, *params):
        L.require_cuda(ws)
        # one row broadcast over the style inputs (broadcast_rows): read in place through that row, every layer's row index 0
        ctx.bcast = None
        if ws.dim() == 3 and ws.shape[0] == 1 and ws.shape[1] > 1 and ws.stride(1) == 0 and ws.stride(2) == 1 and ws.dtype == torch.float32:
            ctx.bcast = tuple(ws.shape)
            ws = ws[:, :1, :]
        ws = ws.contiguous().float()
        N = ws.shape[0]
        layers, wsqs, pi = [], [], 0
        for wrow, wgain, bgain, post, has_bias, wsq in plan:
            w = params[pi].detach().contiguous().float()
            b = params[pi + 1].detach().contiguous().float() if has_bias else None
            pi += 2 if has_bias else 1
            layers.append((w, b, 0 if ctx.bcast is not None else wrow, wgain, bgain, post))
            wsqs.append(wsq)
        outs = tuple(torch.empty((N, ly[0].shape[0]), device=ws.device) for ly in layers)
        ds = [torch.empty((N, q.shape[0]), device=ws.device) if q is not None else None for q in wsqs]
        H.style_affine(ws, layers, outs=outs, demod=[(q, d, None, None) if q is not None else None for q, d in zip(wsqs, ds)])
        ctx.layers, ctx.ws, ctx.wsqs = layers, ws, wsqs
        ctx.save_for_backward(*outs, *[d for d in ds if d is not None])
        return outs + tuple(d for d in ds if d is not None)

    @staticmethod
    def backward(ctx, *grads):
        nl = len(ctx.layers)
        saved = ctx.saved_tensors
        outs, dsaved = saved[:nl], list(saved[nl:])
        dev = ctx.ws.device
        # which affine parameters want a gradient (pivotal tuning): needs_input_grad is aligned with (ws, plan, *params)
        want, pi = [], 2
        for ly in ctx.layers:
            ww = ctx.needs_input_grad[pi]
            wb = ly[1] is not None and ctx.needs_input_grad[pi + 1]
            want.append((ww, wb))
            pi += 2 if ly[1] is not None else 1
        any_w = any(a or b for a, b in want)
        dws = None
        if not (ctx.needs_input_grad[0] or any_w):
            return (None, None) + (None,) * (pi - 2)
        if ctx.needs_input_grad[0]:
            # (a broadcast row: the whole gradient lands in row 0 of the zero-filled [1,L,D] tensor -- BroadcastRowsFn takes it from there)
            dws = H.zeros(ctx.ws.shape if ctx.bcast is None else ctx.bcast, dev)
        douts = [g.contiguous().float() if g is not None else None for g in grads[:nl]]
        dds = [g.contiguous().float() if g is not None else None for g in grads[nl:]]
        demod, di = [], 0
        extra_shapes = [tuple(o.shape) if q is not None else None for o, q in zip(outs, ctx.wsqs)]
        extras = _zeros_views(dev, *extra_shapes)
        for q, ex in zip(ctx.wsqs, extras):
            if q is None:
                demod.append(None)
            else:
                demod.append((q, dsaved[di], dds[di], ex))
                di += 1
        dW = [torch.empty_like(ly[0]) if (w_ and g is not None) else None for ly, (w_, _), g in zip(ctx.layers, want, douts)]
        dB = [torch.empty_like(ly[1]) if (b_ and g is not None) else None for ly, (_, b_), g in zip(ctx.layers, want, douts)]
        H.style_affine(ctx.ws, ctx.layers, outs=outs, douts=douts, dws=dws, demod=demod, backward=True, dweights=dW if any_w else None,
                       dbiases=dB if any_w else None)
        pg = []
        for ly, w_, b_ in zip(ctx.layers, dW, dB):
            pg.append(w_)
            if ly[1] is not None:
                pg.append(b_)
        if dws is not None and ctx.bcast is not None:
            dws._eg3d_row0_total = True
        return (dws, None) + tuple(pg)


def style_bank(ws, entries):
    """entries: list of (FullyConnectedLayer affine, ws row index, post scale, conv layer | None).  Returns (styles, demods) -- two
    lists aligned with `entries` (demods[i] is None for layers without demodulation) -- or None when
    the bank does not apply (non-linear activation, too many layers)."""
    if len(entries) > L.STYLE_BANK_MAX or not ws.is_cuda:
        return None
    plan, params = [], []
    for fc, wrow, post, conv in entries:
        if fc.activation != 'linear' or fc.weight.dtype != torch.float32:
            return None
        # demodulation coefficients of every conv layer from one launch, d d / d styles in the bank's backward; for trainable conv weights
        # the layer adds d d / d weight itself (hipops.weight_grad_finish)
        wsq = conv._cache.get(conv.weight)[2] if conv is not None else None
        plan.append((int(wrow), float(fc.weight_gain), float(fc.bias_gain), float(post), fc.bias is not None, wsq))
        params.append(fc.weight)
        if fc.bias is not None:
            params.append(fc.bias)
    res = list(StyleBankFn.apply(ws, tuple(plan), *params))
    styles, rest = res[:len(entries)], res[len(entries):]
    demods, di = [], 0
    for e, pl in zip(entries, plan):
        if pl[5] is not None:
            demods.append(rest[di])
            di += 1
        else:
            demods.append(None)
    return styles, demods


class ToRGBFn(torch.autograd.Function):
    """y = clamp(conv1x1(x * styles, W) + bias);  out = skip + y (skip optional).  Small channel counts are padded to 4."""

    @staticmethod
    def forward(ctx, x, weight, styles, bias, skip, clamp, cache, want_wgrad, passthrough=False, input_is_layer_output=False, skip_up=False):
        """skip_up=True: `skip` is the HALF-resolution skip image; what is added is upsample2d(skip, [1,3,3,1]) -- inside the conv's epilogue
        where that is possible (no clamp, even size), by UpsampleImgFn's kernel otherwise.
        passthrough=True additionally returns x itself: the consumer of that output (the next block's conv0) then sends its
        gradient through THIS backward, where it is added inside the data-gradient epilogue instead of by a separate autograd add
        (3 x tensor bytes per block, 67 MB tensors in the SR head)."""
        L.require_cuda(x, weight, styles)
        PENDING_DGRAD.clear()            # (entries live inside one backward pass; anything left over belongs to a pass that was abandoned)
        pend = x.__dict__.pop('_eg3d_pending_epi', None) if hasattr(x, '__dict__') else None      # the producing layer's finishing pass is still due (ModConvLayerFn defer_epilogue)
        pre_y = x.__dict__.pop('_eg3d_rgb_y', None) if hasattr(x, '__dict__') else None            # ... or this very layer already ran in the producing launch's epilogue
        if pre_y is not None and (pre_y[7] != x._version or pre_y[4] != styles._version):          # x or the styles were written in place since (a hook, a noise injection): y is stale
            pre_y = None
        if pend is not None and not (H.is_cl(x) and x.dtype == torch.float32 and pend.out is x):
            pend.run()
            pend = None
        x = H.to_cl(x.float())
        styles_key = styles.data_ptr()          # (of the tensor the caller holds: what ModConvLayerFn recorded with a pre-computed y)
        styles = styles.contiguous().float()
        N, Ci, Hh, Ww = x.shape
        Co = weight.shape[0]
        Cp = (Co + 3) // 4 * 4
        clampv = -1.0 if clamp is None else float(clamp)
        if Cp != Co:              # compute all Cp channels: zero weight rows / bias give 0 (+ skip) in the padding channels
            wf, _, b = cache.get_padded(weight, Cp, bias)
        else:
            wf, _, _ = cache.get(weight)
            b = bias.contiguous().float() if bias is not None else None
        cls = H.classes_corr(Hh, Ww, 1, 1, 0)
        y = None
        up_taps = None
        lazy_up = False          # clamped layer (its backward needs y itself): out = upsample2d(skip) + y in ONE pass once y exists (eg3d_upfirdn2d_nhwc_add)
        if skip is not None:
            skip = H.to_cl(skip.float())
            assert skip.shape[1] == Cp
            if skip_up:
                if FUSE_SKIP_UP and clampv < 0 and Hh % 2 == 0 and Ww % 2 == 0 and tuple(skip.shape[2:]) == (Hh // 2, Ww // 2) and (N == 1 or (Hh * Ww) % 128 == 0):
                    up_taps = (0.25, 0.75, 0.75, 0.25)          # [1,3,3,1] / 8, times the per-axis gain 2
                elif FUSE_SKIP_ADD and clampv >= 0 and tuple(skip.shape) == (N, Cp, Hh // 2, Ww // 2) and Hh % 2 == 0 and Ww % 2 == 0:
                    lazy_up = True
                else:
                    skip = H.upfirdn2d_nhwc(skip, fir44(skip.device), up=2, pad=(2, 1, 2, 1), gain=4.0)

        def add_skip(y_):
            if skip is None:
                return y_
            if lazy_up:
                return H.upfirdn2d_nhwc(skip, fir44(skip.device), up=2, pad=(2, 1, 2, 1), gain=4.0, addend=y_)
            return y_ + skip
        # small pixel counts (the 4^2 .. 64^2 blocks): the implicit GEMM's 32-step contraction is all latency there (~21 us for 64 pixels);
        # csrc/torgb_small.hip does the same arithmetic (exact fp32 products) in one short launch
        small = (H.TORGB_SMALL and (N * Hh * Ww <= H.TORGB_SMALL_MAX_PIX or (H.TORGB_MID and pend is None)) and Ci % 8 == 0 and Cp % 32 == 0 and wf.stride(1) == 1
                 and (skip is None or up_taps is not None or lazy_up or tuple(skip.shape) == (N, Cp, Hh, Ww)))
        if pend is not None and not small:
            pend.run()
            pend = None
        if small and clampv < 0 and skip is not None:
            out = H.empty_cl(N, Cp, Hh, Ww, x.device)
            small = H.torgb_small(x, wf, styles, out, bias=b, clamp=-1.0, addend=skip, addend_up2_taps=up_taps, pre=pend)
            if not small and pend is not None:
                pend.run()
            pend = None
        elif small:
            y = H.empty_cl(N, Cp, Hh, Ww, x.device)
            if up_taps is not None:          # clamped layer: its backward needs y itself, the skip image is added by a pass of its own
                skip = H.upfirdn2d_nhwc(skip, fir44(skip.device), up=2, pad=(2, 1, 2, 1), gain=4.0)
                up_taps = None
            small = H.torgb_small(x, wf, styles, y, bias=b, clamp=clampv, pre=pend)
            if not small and pend is not None:
                pend.run()
            pend = None
            if small:
                out = add_skip(y)
        if small:
            pass
        elif clampv < 0 and skip is not None:
            out = H.empty_cl(N, Cp, Hh, Ww, x.device)
            try:
                H.conv_igemm(x, wf, Ci, Cp, out, cls, in_scale=styles, epi=L.EPI_FWD, bias=b, act='linear', gain=1.0, clamp=-1.0, addend=skip,
                             precision=_igemm_precision(), addend_up2_taps=up_taps)
            except RuntimeError:
                if up_taps is None:
                    raise
                # the launch could not take the half-resolution image (tile / alignment conditions of the vector epilogue): up-sample first
                skip = H.upfirdn2d_nhwc(skip, fir44(skip.device), up=2, pad=(2, 1, 2, 1), gain=4.0)
                H.conv_igemm(x, wf, Ci, Cp, out, cls, in_scale=styles, epi=L.EPI_FWD, bias=b, act='linear', gain=1.0, clamp=-1.0, addend=skip,
                             precision=_igemm_precision())
        else:
            if (pre_y is not None and Cp == 4 and tuple(pre_y[0].shape) == (N, Cp, Hh, Ww) and pre_y[1:3] == (weight.data_ptr(), weight._version)
                    and pre_y[3] == styles_key and pre_y[5] == (None if bias is None else (bias.data_ptr(), bias._version)) and pre_y[6] == clampv):
                y = pre_y[0]                # evaluated by conv1's epilogue from the values it was writing: no launch here
            else:
                y = H.empty_cl(N, Cp, Hh, Ww, x.device)
                H.conv_igemm(x, wf, Ci, Cp, y, cls, in_scale=styles, epi=L.EPI_FWD, bias=b, act='linear', gain=1.0, clamp=clampv,
                             precision=_igemm_precision())
            out = add_skip(y)
        ctx.save_for_backward(x, weight, styles, y if clampv >= 0 else None)
        ctx.cfg = (clampv, cache, want_wgrad, Cp, skip is not None, bool(skip_up))
        ctx.fuse_input = bool(input_is_layer_output)         # x is conv1's output handed over directly (see ModConvLayerFn)
        if not passthrough:
            return out
        xv = x.view_as(x)
        if getattr(x, '_eg3d_amax', None) is not None:
            H.tag_amax(xv, x._eg3d_amax)
        return out, xv

    @staticmethod
    def backward(ctx, dout, dx_pass=None):
        x, weight, styles, y = ctx.saved_tensors
        clampv, cache, want_wgrad, Cp, has_skip, skip_up = ctx.cfg
        need_x, need_w, need_s, need_b, need_skip = ctx.needs_input_grad[:5]
        need_w = need_w and want_wgrad
        dout = H.to_cl(dout.float())
        N, Ci, Hh, Ww = x.shape
        Co = weight.shape[0]
        dev = x.device
        dy = dout
        dbias = None
        dy_amax = None
        if clampv >= 0 or need_b:
            dbias_p = H.zeros((Cp,), dev) if need_b else None
            dy = H.empty_cl(N, Cp, Hh, Ww, dev)
            if (need_w and _igemm_precision() == 'f16x3') or (SPLIT_DZ and Cp == 4):
                dy_amax = H.zeros((1,), dev)          # max|dy| from the same pass: the weight gradient can then run in the two-piece fp16 arithmetic
                                                      # (and the fused split of the producing layer's dz takes its range bound from it)
            H.epilogue_bwd(dout, y if y is not None else dout, dy, act='linear', gain=1.0, clamp=clampv, dbias=dbias_p, dz_amax=dy_amax)
            dbias = dbias_p[:Co] if need_b else None
        dx = ds = None
        pdg = PENDING_DGRAD.pop(x.data_ptr(), None)          # the pass-through gradient is an unfinished split-K sum (ModConvLayerFn.backward)
        if pdg is not None and (dx_pass is None or dx_pass.data_ptr() != pdg[0].data_ptr() or tuple(dx_pass.shape) != tuple(x.shape)):
            raise RuntimeError('ToRGBFn.backward: a deferred data-gradient finish is pending for this input but the gradient that arrived is another tensor '
                               '(set EG3D_DEFER_DGRAD_FINISH=0 and report the configuration)')
        if need_x or need_s:
            wa_p = cache.get(weight)[1] if Cp == Co else cache.get_padded(weight, Cp)[1]    # contraction dim (output channels) padded to 4
            dx = H.empty_cl(N, Ci, Hh, Ww, dev)
            ds = H.zeros((N, Ci), dev)
            small_ok = H.TORGB_SMALL and (N * Hh * Ww <= H.TORGB_SMALL_BWD_MAX_PIX or H.TORGB_MID_BWD) and Ci % 32 == 0 and Cp % 8 == 0 and wa_p.stride(1) == 1
            if pdg is not None and not (small_ok and H.is_cl(dx_pass) and dx_pass.dtype == torch.float32):
                dx_pass, pdg = _finish_pending(pdg, x), None
            add = H.to_cl(dx_pass.float()) if dx_pass is not None else None
            prod, spec, pacc = _act_bwd_for(x, dev) if (need_x and ctx.fuse_input) else (None, None, None)      # x = conv1's output: run its activation backward here
            fkw = dict(act_bwd=spec, out_amax=pacc[4]) if prod is not None else {}
            if pdg is not None and prod is not None and Cp == 4 and Ci % 4 == 0 and Ci <= 1024 and TORGB4_ELEMENTWISE:
                add, pdg = _finish_pending(pdg, x), None
            if prod is not None and Cp == 4 and Ci % 4 == 0 and Ci <= 1024 and TORGB4_ELEMENTWISE:
                # four outputs (the SR head's toRGB): the data gradient is four multiply-adds per element -- one element-wise pass with the
                # producing layer's activation backward instead of a GEMM launch with a 4-deep contraction (105 -> 60 us at 512^2 x 128)
                add_amax = None
                if add is not None:       # the report belongs to THIS tensor object (an address can be recycled; a copy made by autograd has no report)
                    ent = _DX_AMAX.pop(add.data_ptr(), None)
                    if ent is not None and (ent[1]() is add or ent[1]() is dx_pass):
                        add_amax = ent[0]
                if getattr(prod, 'split_ok', False) and dy_amax is not None and (add is None or add_amax is not None) and Ci % 8 == 0:
                    # the producing layer's backward only feeds dz to the pre-split data gradient: write its operand image here (no split pass,
                    # no fp32 dz -- a stride-0 token stands in for the gradient tensor autograd passes on)
                    simg = H.torgb_dgrad_act_split(dy, wa_p, x, styles, spec, dy_amax, ds=ds, addend=add, addend_amax=add_amax)
                    dx = _dz_token(x.shape, dev)
                    prod.fused = (dx,) + tuple(pacc) + (simg,)
                    did = None
                else:
                    H.torgb_dgrad_act(dy, wa_p, x, styles, dx, spec, ds=ds, addend=add, dz_amax=pacc[4])
                    did = True
            else:
                did = None
                if small_ok:
                    # small pixel counts: the low-latency launch (csrc/torgb_small.hip), same epilogue as the implicit GEMM's
                    did = H.torgb_small_bwd(dy, wa_p, styles, x, dx, ds=ds, addend=add, **fkw,
                                            **(dict(addend_scale=pdg[1], addend_ds=pdg[2]) if pdg is not None else {}))
                if did is None and pdg is not None:
                    add, pdg = _finish_pending(pdg, x), None
                if did is None:
                    did = H.conv_igemm(dy, wa_p, Cp, Ci, dx, H.classes_corr_adjoint(Hh, Ww, 1, 1, 0), epi=L.EPI_BWD, out_scale=styles, xin=x, ds=ds, addend=add,
                                       **fkw)
            if prod is not None and did is True:
                prod.fused = (dx,) + tuple(pacc)
        elif dx_pass is not None:
            dx = _finish_pending(pdg, x) if pdg is not None else dx_pass
        dweight = None
        if need_w:
            dwp = H.zeros((Co, Ci), dev)
            # a 1x1 conv with 3..96 outputs: on the fp32 matrix path the 128 x 128 tile costs 64 MFMAs of 64 cycles per 32 cells whatever the
            # channel count; the split-fp16 kernel needs 24 of 32 (same fp32-equivalent arithmetic as the 3x3 layers)
            if dy_amax is not None:
                H.conv_wgrad(x, dy, Ci, Co, dwp, H.classes_corr(Hh, Ww, 1, 1, 0), in_scale=styles, precision='f16x3', g_amax=dy_amax)
            else:
                H.conv_wgrad(x, dy, Ci, Co, dwp, H.classes_corr(Hh, Ww, 1, 1, 0), in_scale=styles)
            if H.DEFERRED_CONV_WGRADS is not None:
                # inside hipops.deferred_weight_grads() the GEMM above is only QUEUED: dwp is still zero here.  Handing its view to autograd was
                # right only while AccumulateGrad stole the tensor; with an existing .grad (zero_grad(set_to_none=False), accumulation, a hook) the
                # toRGB weight gradient was silently zero.  Same route as the 3x3 layers: the flush sets / accumulates weight.grad.
                dweight = H.weight_grad_finish(dwp, weight, None, None, None)
            else:
                dweight = dwp.view(Co, Ci, 1, 1)
        dskip = None
        if need_skip and has_skip:        # the adjoint of the up-sampling when the node took the half-resolution image
            dskip = H.upfirdn2d_nhwc(dout, fir44(dev), down=2, pad=(1, 1, 1, 1), flip=True, gain=4.0) if skip_up else dout
        return (dx if need_x else None, dweight, ds if need_s else None, dbias, dskip, None, None, None, None, None, None)


class _SliceRgb4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, res, share):
        L.require_cuda(feat)
        feat = feat.contiguous().float()
        n, r, c = feat.shape
        y = torch.empty((n, res, res, 4), device=feat.device)
        L.check(L.lib().eg3d_slice_rgb4_fwd(feat.data_ptr(), y.data_ptr(), n * r, c, L.stream_ptr()), 'slice_rgb4_fwd')
        ctx.shape = (n, r, c)
        if share:           # the feature image's other consumer (the SR head) reads it through THIS node: both gradients arrive in one backward call
            return y.permute(0, 3, 1, 2), feat.view_as(feat)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g, g_feat=None):
        n, r, c = ctx.shape
        dev = (g if g is not None else g_feat).device
        if g is None:
            return g_feat, None, None
        g = H.to_cl(g.float())
        dx = torch.empty((n, r, c), device=dev)
        if g_feat is not None:      # dx = g_feat + (g.xyz, 0, ...) in one pass (eg3d_slice_rgb4_bwd_add) instead of a scatter pass and autograd's add
            g_feat = g_feat.contiguous().float()
            L.check(L.lib().eg3d_slice_rgb4_bwd_add(g.data_ptr(), g_feat.data_ptr(), dx.data_ptr(), n * r, c, L.stream_ptr()), 'slice_rgb4_bwd_add')
        else:
            L.check(L.lib().eg3d_slice_rgb4_bwd(g.data_ptr(), dx.data_ptr(), n * r, c, L.stream_ptr()), 'slice_rgb4_bwd')
        return dx, None, None


def slice_rgb4(feat: torch.Tensor, res: int, share: bool = False):
    """[N, res*res, C] rendered features -> the raw RGB image [N,4,res,res] channels_last with 4-float pixels (features[:, :3], channel 3 = 0;
    triplane.py:84-85) in one launch per direction.  share=True: returns (image, features) where `features` is the input itself behind this
    node -- a caller that hands it to the feature image's other consumer gets the two gradients summed inside the backward launch."""
    return _SliceRgb4Fn.apply(feat, int(res), bool(share))


class UpsampleImgFn(torch.autograd.Function):
    """upfirdn2d.upsample2d(img, [1,3,3,1]) on a channels_last image (C % 4 == 0)."""

    @staticmethod
    def forward(ctx, img):
        img = H.to_cl(img.float())
        return H.upfirdn2d_nhwc(img, fir44(img.device), up=2, pad=(2, 1, 2, 1), gain=4.0)

    @staticmethod
    def backward(ctx, g):
        g = H.to_cl(g.float())
        return H.upfirdn2d_nhwc(g, fir44(g.device), down=2, pad=(1, 1, 1, 1), flip=True, gain=4.0)


class RayGenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c2w, K, res):
        L.require_cuda(c2w, K)
        c2w = c2w.contiguous().float()
        K = K.contiguous().float()
        o, d = H.ray_gen_fwd(c2w, K, res)
        ctx.save_for_backward(c2w, K)
        ctx.res = res
        return o, d

    @staticmethod
    def backward(ctx, g_o, g_d):
        c2w, K = ctx.saved_tensors
        g_o = g_o.contiguous().float() if g_o is not None else None
        g_d = g_d.contiguous().float() if g_d is not None else None
        d_c2w, d_K = H.ray_gen_bwd(c2w, K, g_o, g_d, ctx.res, want_K=ctx.needs_input_grad[1])
        return d_c2w, d_K, None


SAMPLER_DEBUG = None          # a dict while a parity test wants the sampler's indices (tests/test_gpu_ops.py::test_sampler_indices_exact)


class RenderFn(torch.autograd.Function):
    """ImportanceRenderer.forward fused.  planes: CL [N,96,Hp,Wp]; returns rgb [N,R,32], depth [N,R,1], wsum [N,R,1]."""

    @staticmethod
    def forward(ctx, planes, origins, dirs, w0, b0, w1, b1, u1, u2, opts, lr_mul, ray_limits):
        L.require_cuda(planes, origins, dirs, w0)
        planes = H.to_cl(planes.float())
        origins = origins.contiguous().float()
        dirs = dirs.contiguous().float()
        N, R = origins.shape[0], origins.shape[1]
        dev = planes.device
        Dc, Df = int(opts['depth_resolution']), int(opts['depth_resolution_importance'])
        hid, cin = w0.shape
        g0, g1 = lr_mul / math.sqrt(cin), lr_mul / math.sqrt(hid)
        w0g, b0g, w1t, b1g = H.memo(('decoder', lr_mul), [w0, b0, w1, b1], lambda: (
            (w0.detach().float() * g0).contiguous(), (b0.detach().float() * lr_mul).contiguous(),
            (w1.detach().float() * g1).t().contiguous(), (b1.detach().float() * lr_mul).contiguous()))
        u1 = u1.contiguous().float()
        u2 = u2.contiguous().float() if u2 is not None else None
        rgb = torch.empty((N, R, w1.shape[0] - 1), device=dev)
        depth = torch.empty((N, R, 1), device=dev)
        wsum = torch.empty((N, R, 1), device=dev)
        fine = torch.empty((N, R, max(Df, 1)), device=dev)
        rl = ray_limits.contiguous().float() if ray_limits is not None else None
        save = None
        if any(ctx.needs_input_grad[:7]):          # training mode: keep (sigma, colour) per sample (207 MB at the FFHQ config)
            S = N * R * 2 * max(Dc, Df)
            save = (torch.empty((S,), device=dev), torch.empty((S, w1.shape[0] - 1), device=dev))
        pos = None
        rows = save
        if rows is None and Df > 0 and RENDER_PIPELINE and RENDER_PIPELINE_NOGRAD and N * R >= 4096:
            # no-grad rendering (orbit frames, the canonical view of the warping loss, evaluation): the same pipeline with the per-sample rows
            # as scratch -- since the decoder moved to the 16-bit matrix cores it beats the fused per-ray kernel (0.50 vs 0.84 ms at 128^2 x 96)
            S = N * R * 2 * max(Dc, Df)
            rows = (torch.empty((S,), device=dev), torch.empty((S, w1.shape[0] - 1), device=dev))
        feat = None
        if rows is not None and Df > 0 and RENDER_PIPELINE:      # sample-level decode on the matrix cores between ray-level stages
            pos = torch.empty((2, N * R, max(Dc, Df), 4), device=dev)
            if RENDER_FEAT_ROWS:        # the tri-plane gather as its own pass; the backward re-reads the rows instead of gathering again
                feat = torch.empty((N * R * 2 * max(Dc, Df), 32), device=dev)
        if pos is not None:
            minmax = torch.empty((2,), device=dev)      # the pipelined forward's first launch writes (+inf, -inf) itself
        else:
            minmax = _minmax_init(dev) + 0.0            # device-side copy by an elementwise kernel: no host transfer (graph-capturable) and no
                                                        # memcpy node (those split a captured graph into separately submitted segments)
        dbg = None
        if SAMPLER_DEBUG is not None and Df > 0:         # parity tests: the integer side of the sampler (eg3d_render_params::dbg_*)
            dbg = (torch.full((N * R, Df, 3), -1, dtype=torch.int32, device=dev), torch.full((N * R, Dc + Df), -1, dtype=torch.int32, device=dev),
                   torch.zeros((N * R, max(Dc - 3, 1)), device=dev))
            SAMPLER_DEBUG.update(inds=dbg[0], ranks=dbg[1], cdf=dbg[2], fine=fine, rows=rows if pos is not None else save, pos=pos)
        p = H.make_render_params(planes, origins, dirs, u1, u2, opts, w0g, b0g, w1t, b1g, rgb, depth, wsum, minmax, fine, rl, rows if pos is not None else save,
                                 pos_rows=pos, feat_rows=feat, dbg=dbg)
        with H._Span('render_fwd'):
            H.render_fwd(p)
            H.render_finalize(depth, minmax)
        ctx.has_feat = feat is not None and save is not None
        ctx.set_materialize_grads(False)       # an unused output (depth, weight sum) arrives as None in backward instead of a freshly filled zero tensor
        ctx.save_for_backward(planes, origins, dirs, w0g, b0g, w1t, b1g, u1, u2, minmax, fine, rl, *(save or ()), *((feat,) if ctx.has_feat else ()))
        ctx.cfg = (dict(opts), g0, g1, lr_mul)
        return rgb, depth, wsum

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_wsum):
        planes, origins, dirs, w0g, b0g, w1t, b1g, u1, u2, minmax, fine, rl = ctx.saved_tensors[:12]
        save = ctx.saved_tensors[12:14]
        feat_rows = ctx.saved_tensors[14] if ctx.has_feat else None
        opts, g0, g1, lr_mul = ctx.cfg
        need = ctx.needs_input_grad
        dev = planes.device
        N, R = origins.shape[0], origins.shape[1]
        g_rgb = g_rgb.contiguous().float() if g_rgb is not None else torch.zeros((N, R, w1t.shape[1] - 1), device=dev)
        g_depth = g_depth.contiguous().float() if g_depth is not None else None
        g_wsum = g_wsum.contiguous().float() if g_wsum is not None else None
        p = H.make_render_params(planes, origins, dirs, u1, u2, opts, w0g, b0g, w1t, b1g, None, None, None, minmax, fine, rl, save, feat_rows=feat_rows)
        d_planes = H.zeros_cl(*planes.shape, dev) if need[0] else None
        d_o = torch.empty_like(origins) if (need[1] or need[2]) else None
        d_d = torch.empty_like(dirs) if (need[1] or need[2]) else None
        dumps = gram = None
        if any(need[3:7]) and GRAM_FUSED and feat_rows is not None and torch.is_tensor(g0) is False:
            # decoder-weight gradients contracted inside the sample-level kernel (eg3d_render_bwd_params::gram_*): no operand dumps, no GEMM passes
            gram = (H.zeros((64, 32), dev), H.zeros((64,), dev), H.zeros((33, 64), dev), H.zeros((33,), dev), g0, g1, lr_mul)
        elif any(need[3:7]):
            D = max(p.Dc, p.Df)
            S = N * R * 2 * D
            # with equal coarse / fine counts every row is a live sample and is written by the sample-level kernel
            mk = torch.empty if p.Dc == p.Df else torch.zeros
            # (the feature operand of the decoder-weight Gram product is the saved feature rows themselves when the forward kept them; rows of
            #  absent samples hold zeros there as well)
            dumps = [mk((S, 64), device=dev), mk((S, 64), device=dev), mk((S, 33), device=dev), feat_rows if feat_rows is not None else mk((S, 32), device=dev)]
        with H._Span('render_bwd'):
            H.render_bwd(p, g_rgb, g_depth, g_wsum, d_planes, d_o, d_d, dumps, gram)
        dw0 = db0 = dw1 = db1 = None
        if gram is not None:
            dw0, db0, dw1, db1 = gram[:4]
        elif dumps is not None:
            dpre, hid, dout, feat = dumps
            dw0, db0 = H.rows_gram(dpre, feat, g0, lr_mul)          # [64,32] = g0 dpre^T feat and lr_mul x its column sums over 1.57 M samples
            dw1, db1 = H.rows_gram(dout, hid, g1, lr_mul)           # [33,64]  (the runtime gains applied to the partial sums: no scaling passes)
        return (d_planes, d_o if need[1] else None, d_d if need[2] else None, dw0, db0, dw1, db1, None, None, None, None, None)
