"""One-shot latent initialiser of the projector on the gfx950 kernels (second half).

The reference starts Phase A at `w_avg + e4e_enc(target)` (training/projectors/w_projector.py:70-74,100,118): `e4e_enc` is
`pSp2` (models/e4e/psp.py:18-65) wrapping `Encoder4Editing(50, 'ir_se')` (models/e4e/encoders/psp_encoders.py:124-200): an IR-SE-50
trunk (models/e4e/encoders/helpers.py:22-120: 24 bottleneck_IR_SE units, PReLU, squeeze-excite) with an FPN and 18 `GradualStyleBlock`
heads (:35-57); pSp2.forward returns codes[:, 0] -- the output of head 0 on the coarsest feature map.  `Encoder4Editing` / `PSPEncoder`
here have the same module trees and state-dict keys (`e4e_ffhq.pt` / `e4e_afhq.pt` load with `load_state_dict`, keys `encoder.*`).

Evaluation (inference only, run once per image): every convolution -- 3x3 stride 1 / 2, the 1x1 shortcuts, the squeeze-excite 1x1s --
goes through the implicit-GEMM kernel (loss_nets.conv_act); a BatchNorm that FOLLOWS a convolution is folded into its weights and
epilogue bias, one that precedes it (the units open with BatchNorm before a zero-padded conv, which does not commute with folding) is
an element-wise affine map; PReLU, sigmoid gating and the bilinear FPN up-sampling are element-wise device ops."""
import math

import torch
import torch.nn.functional as F

from . import hipops as H
from .loss_nets import conv_act

IRSE50 = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))       # (in, depth, units) per stage; first unit of a stage has stride 2


def _bn_affine(bn):
    a = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return a, bn.bias - bn.running_mean * a


def _bn(x, bn):
    a, b = _bn_affine(bn)
    return H.to_cl(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))


def _conv(x, conv, act='linear', bn=None):
    """conv (+ folded eval-mode BatchNorm that follows it) (+ bias) on the implicit-GEMM kernel; channels padded to multiples of 4."""
    w, b = conv.weight, conv.bias
    co = w.shape[0]
    if bn is not None:
        a, sh = _bn_affine(bn)
        w = w * a.view(-1, 1, 1, 1)
        b = sh if b is None else b * a + sh
    if b is None:
        b = w.new_zeros(co)
    cop = (co + 3) // 4 * 4
    if cop != co:
        w = torch.cat([w, w.new_zeros(cop - co, *w.shape[1:])], 0)
        b = torch.cat([b, b.new_zeros(cop - co)])
    return conv_act(x, w, b, conv.stride[0], conv.padding[0], act)


class SEModule(torch.nn.Module):
    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = torch.nn.Conv2d(channels, channels // reduction, 1, bias=False)
        self.fc2 = torch.nn.Conv2d(channels // reduction, channels, 1, bias=False)

    def forward(self, x):
        s = H.to_cl(x.mean((2, 3), keepdim=True))
        s = _conv(_conv(s, self.fc1, 'relu'), self.fc2)
        return x * torch.sigmoid(s[:, :x.shape[1]])


class bottleneck_IR_SE(torch.nn.Module):
    """helpers.py:101-120: shortcut (stride-s subsampling, or 1x1 conv + BN) + [BN, 3x3, PReLU, 3x3 stride s, BN, SE]."""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.stride = stride
        if in_channel == depth:
            self.shortcut_layer = torch.nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = torch.nn.Sequential(torch.nn.Conv2d(in_channel, depth, 1, stride, bias=False), torch.nn.BatchNorm2d(depth))
        self.res_layer = torch.nn.Sequential(torch.nn.BatchNorm2d(in_channel), torch.nn.Conv2d(in_channel, depth, 3, 1, 1, bias=False),
                                             torch.nn.PReLU(depth), torch.nn.Conv2d(depth, depth, 3, stride, 1, bias=False),
                                             torch.nn.BatchNorm2d(depth), SEModule(depth, 16))

    def forward(self, x):
        r = self.res_layer
        if isinstance(self.shortcut_layer, torch.nn.MaxPool2d):
            sc = x[:, :, ::self.stride, ::self.stride]              # MaxPool2d(kernel 1, stride s)
        else:
            sc = _conv(x, self.shortcut_layer[0], bn=self.shortcut_layer[1])
        y = _conv(_bn(x, r[0]), r[1])
        y = H.to_cl(F.prelu(y, r[2].weight))
        y = r[5](_conv(y, r[3], bn=r[4]))
        return H.to_cl(y + sc)


class EqualLinear(torch.nn.Module):
    """models/e4e/stylegan2/model.py:129-158 without activation: y = x (W / sqrt(in) lr_mul)^T + b lr_mul."""

    def __init__(self, in_dim, out_dim, lr_mul=1):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(out_dim, in_dim) / lr_mul)
        self.bias = torch.nn.Parameter(torch.zeros(out_dim))
        self.scale, self.lr_mul = lr_mul / math.sqrt(in_dim), lr_mul

    def forward(self, x):
        return F.linear(x, self.weight * self.scale, self.bias * self.lr_mul)


class GradualStyleBlock(torch.nn.Module):
    """psp_encoders.py:35-57: log2(spatial) x [3x3 stride-2 conv, LeakyReLU(0.01)] down to 1x1, then an equalised linear layer."""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c, self.spatial = out_c, spatial
        mods = []
        for i in range(int(math.log2(spatial))):
            mods += [torch.nn.Conv2d(in_c if i == 0 else out_c, out_c, 3, 2, 1), torch.nn.LeakyReLU()]
        self.convs = torch.nn.Sequential(*mods)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def forward(self, x):
        for m in self.convs:
            if isinstance(m, torch.nn.Conv2d):
                x = H.to_cl(F.leaky_relu(_conv(x, m), 0.01))           # nn.LeakyReLU() default slope
        return self.linear(x.reshape(-1, self.out_c))


class Encoder4Editing(torch.nn.Module):
    def __init__(self, num_layers=50, mode='ir_se', opts=None):
        super().__init__()
        if num_layers != 50 or mode != 'ir_se':
            raise NotImplementedError('the inversion pipeline uses Encoder4Editing(50, "ir_se") (models/e4e/psp.py:27)')
        self.input_layer = torch.nn.Sequential(torch.nn.Conv2d(3, 64, 3, 1, 1, bias=False), torch.nn.BatchNorm2d(64), torch.nn.PReLU(64))
        units = []
        for cin, depth, n in IRSE50:
            units += [bottleneck_IR_SE(cin, depth, 2)] + [bottleneck_IR_SE(depth, depth, 1) for _ in range(n - 1)]
        self.body = torch.nn.Sequential(*units)
        self.style_count, self.coarse_ind, self.middle_ind = 18, 3, 7
        self.styles = torch.nn.ModuleList(GradualStyleBlock(512, 512, 16 if i < 3 else (32 if i < 7 else 64)) for i in range(18))
        self.latlayer1 = torch.nn.Conv2d(256, 512, 1)
        self.latlayer2 = torch.nn.Conv2d(128, 512, 1)
        self.eval()

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('Encoder4Editing runs with frozen BatchNorm statistics (w_projector.py:70 .eval())')
        return super().train(False)

    def trunk(self, x):
        n, c, h, w = x.shape
        x = torch.cat([x.float(), x.new_zeros(n, 1, h, w, dtype=torch.float32)], 1).contiguous(memory_format=torch.channels_last)
        x = _conv(x, self.input_layer[0], bn=self.input_layer[1])
        x = H.to_cl(F.prelu(x, self.input_layer[2].weight))
        feats = {}
        for i, unit in enumerate(self.body):
            x = unit(x)
            if i in (6, 20, 23):
                feats[i] = x
        return feats[6], feats[20], feats[23]

    @torch.no_grad()
    def forward(self, x, first_only=False):
        """[N,3,H,W] -> codes [N,18,512] (psp_encoders.py:171-200, progressive stage = Inference); first_only: only head 0, [N,512]."""
        c1, c2, c3 = self.trunk(x)
        w0 = self.styles[0](c3)
        if first_only:
            return w0
        codes = [w0.clone() for _ in range(self.style_count)]
        feats = c3
        for i in range(1, self.style_count):
            if i == self.coarse_ind:
                p2 = H.to_cl(F.interpolate(c3, size=c2.shape[-2:], mode='bilinear', align_corners=True) + _conv(c2, self.latlayer1))
                feats = p2
            elif i == self.middle_ind:
                feats = H.to_cl(F.interpolate(p2, size=c1.shape[-2:], mode='bilinear', align_corners=True) + _conv(c1, self.latlayer2))
            codes[i] = codes[i] + self.styles[i](feats)
        return torch.stack(codes, 1)


class PSPEncoder(torch.nn.Module):
    """pSp2 (models/e4e/psp.py): forward(img) = encoder(img)[:, 0] -- the latent offset the projector adds to w_avg."""

    def __init__(self):
        super().__init__()
        self.encoder = Encoder4Editing(50, 'ir_se')

    @torch.no_grad()
    def forward(self, x):
        return self.encoder(x, first_only=True)
