"""Super-resolution head (reference: training/superresolution.py:262-290, SuperresolutionHybrid8XDC).
Two SynthesisBlocks 32->256 @256^2 and 256->128 @512^2 on the fused gfx950 ops; the 3-channel image is carried padded
to 4 channels (16-byte pixels) and sliced at the end."""
import torch

from .. import fused
from .. import hipops as H
from ..reference_binding import ReferenceStateMixin
from .networks_stylegan2 import SynthesisBlock, SynthesisBlockNoUp


class SuperresolutionHybrid8XDC(ReferenceStateMixin, torch.nn.Module):
    # what the five heads of the reference differ in (training/superresolution.py:29-152, 262-290); the subclasses below override these
    block0_up = 2               # 1: the head starts with a SynthesisBlockNoUp
    resize_when = 'ne'          # inputs are resized to input_resolution when their size differs from it ('ne') / is smaller ('lt')
    has_filter_buffer = False   # the head registers a `resample_filter` buffer of its own (state-dict key)

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias=True, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, sr_widths=(256, 128), input_resolution=128, **block_kwargs):
        super().__init__()
        assert img_resolution == input_resolution * self.block0_up * 2
        use_fp16 = sr_num_fp16_res > 0
        self.input_resolution = input_resolution
        self.sr_antialias = sr_antialias
        clamp = 256 if use_fp16 else None
        c0, c1 = sr_widths
        self.block0 = (SynthesisBlock if self.block0_up == 2 else SynthesisBlockNoUp)(
            channels, c0, w_dim=block_kwargs.pop('w_dim', 512), resolution=input_resolution * self.block0_up, img_channels=3, is_last=False,
            use_fp16=use_fp16, conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(c0, c1, w_dim=self.block0.w_dim, resolution=img_resolution, img_channels=3, is_last=True,
                                     use_fp16=use_fp16, conv_clamp=clamp, **block_kwargs)
        if self.has_filter_buffer:
            from ..torch_utils.ops import upfirdn2d
            self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))

    def bank_entries(self, ws):
        """Style-bank entries of the two blocks (all read the LAST row of ws): lets the backbone's bank launch compute them too."""
        last = ws.shape[1] - 1
        return [(fc, last, post, conv) for fc, _, post, conv in self.block0.affine_entries(0) + self.block1.affine_entries(0)]

    def forward(self, rgb, x, ws, noise_inject=None, _bank=None, **block_kwargs):
        ws_all = ws
        if (x.shape[-1] != self.input_resolution) if self.resize_when == 'ne' else (x.shape[-1] < self.input_resolution):
            size = (self.input_resolution, self.input_resolution)
            x = torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
            rgb = torch.nn.functional.interpolate(rgb, size=size, mode='bilinear', align_corners=False, antialias=self.sr_antialias)
        n, _, h, w = rgb.shape
        rgb4 = getattr(rgb, '_eg3d_padded4', None)          # set by TriPlaneGenerator.synthesis: sliced and padded in one launch
        if rgb4 is None:
            rgb4 = torch.cat([rgb, rgb.new_zeros(n, 1, h, w)], 1).contiguous(memory_format=torch.channels_last)
        # both blocks read rows 0..2 of ws[:, -1:, :].repeat(1, 3, 1) (superresolution.py:63-64), i.e. the LAST row of ws six times: the bank
        # reads that row in place (no repeat / slice copies and their backward), the repeated tensor is only built for the per-layer path
        if ws_all.is_cuda and self.block0.conv1.weight.requires_grad:
            fused.prepack_weights(self.block0.packed_layers() + self.block1.packed_layers())
        e0, e1 = self.block0.affine_entries(0), self.block1.affine_entries(0)
        last = ws_all.shape[1] - 1
        bank = _bank if _bank is not None else fused.style_bank(ws_all.float(), [(fc, last, post, conv) for fc, _, post, conv in e0 + e1])
        ws = ws_all[:, -1:, :].expand(-1, 3, -1) if bank is not None else ws_all[:, -1:, :].repeat(1, 3, 1)
        s0, s1 = ((bank[0][:len(e0)], bank[1][:len(e0)]), (bank[0][len(e0):], bank[1][len(e0):])) if bank is not None else (None, None)
        x, rgb4 = self.block0(x, rgb4, ws, noise_inject=noise_inject, _name='superresolution.block0', styles=s0, **block_kwargs)
        x, rgb4 = self.block1(x, rgb4, ws, noise_inject=noise_inject, _name='superresolution.block1', styles=s1, **block_kwargs)
        img = rgb4[:, :3]                 # a view of the image with 4-float pixels (channel 3 = 0), which the fused loss-side kernels read (inversion.py)
        img._eg3d_padded4 = rgb4
        return img


class SuperresolutionHybrid8X(SuperresolutionHybrid8XDC):
    """128^2 -> 512^2 head with the narrower blocks 32 -> 128 @256^2 and 128 -> 64 @512^2 (reference: training/superresolution.py:29-58);
    everything else as the 8XDC head."""
    has_filter_buffer = True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, **kw):
        kw.setdefault('sr_widths', (128, 64))
        super().__init__(channels, img_resolution, sr_num_fp16_res, sr_antialias, **kw)


class SuperresolutionHybrid4X(SuperresolutionHybrid8XDC):
    """128^2 -> 256^2: SynthesisBlockNoUp 32 -> 128 @128^2, SynthesisBlock 128 -> 64 @256^2; smaller inputs are resized up to 128^2, larger ones
    pass as they are (reference: training/superresolution.py:62-90)."""
    block0_up, resize_when, has_filter_buffer = 1, 'lt', True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, **kw):
        kw.setdefault('sr_widths', (128, 64))
        kw.setdefault('input_resolution', 128)
        super().__init__(channels, img_resolution, sr_num_fp16_res, sr_antialias, **kw)


class SuperresolutionHybrid2X(SuperresolutionHybrid8XDC):
    """64^2 -> 128^2: SynthesisBlockNoUp 32 -> 128 @64^2, SynthesisBlock 128 -> 64 @128^2 (reference: training/superresolution.py:94-122)."""
    block0_up, resize_when, has_filter_buffer = 1, 'ne', True

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, **kw):
        kw.setdefault('sr_widths', (128, 64))
        kw.setdefault('input_resolution', 64)
        super().__init__(channels, img_resolution, sr_num_fp16_res, sr_antialias, **kw)


class SuperresolutionHybridDeepfp32(SuperresolutionHybrid4X):
    """The 4X head of old 256^2 pickles: no `sr_antialias` argument, its resize never anti-aliases (reference: training/superresolution.py:126-152)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias=False, **kw):
        super().__init__(channels, img_resolution, sr_num_fp16_res, False, **kw)
