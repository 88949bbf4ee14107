"""In-loop pose estimator on the gfx950 kernels.

The reference predicts the camera rotation of every latent-projection step with a ResNet-34 whose weights are fine-tuned per image
(scripts/resnet/resnet.py:124-230: torchvision ResNet with a 512 -> 1000 -> 128 -> output_dims head, ReLU between, tanh on the output;
training/projectors/w_projector.py:62 `.eval()` -- BatchNorm uses its running statistics -- :122 Adam over all parameters,
:148-156 quaternion (4) / 6-D (6) / angles (2) output).  `ResNetPose` has the same module tree and state-dict keys, so a checkpoint of
the reference's pose network loads with `load_state_dict`.

Evaluation: every convolution goes through the implicit-GEMM kernel (forward, data gradient, weight gradient).  Eval-mode BatchNorm is
an affine map per channel, y = a x + b with a = gamma / sqrt(var + eps), b = beta - mean a; `a` is folded into the convolution weights
(w' = w a, a tiny tensor op that autograd differentiates into both w and gamma) and `b` is the bias of the conv epilogue, so a
conv + BN + ReLU triple is one launch.  The residual add precedes the ReLU, so the second conv of a block runs with a linear epilogue
and add + ReLU is one more pass.  Max pooling (3x3/2, pad 1) uses eg3d_maxpool2d on a -inf padded image; the three small fully
connected layers are library GEMMs."""
import torch
import torch.nn.functional as F

from . import hipops as H
from .loss_nets import conv_act, conv_scale_act, conv_scale_act_ok, max_pool
from .torch_utils.ops import bias_act


def _bn_affine(bn):
    pre = getattr(bn, '_eg3d_affine', None)        # set for the duration of ResNetPose.forward: all layers' affines from four launches
    if pre is not None:
        return pre
    a = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return a, bn.bias - bn.running_mean * a


def _conv_bn(x, conv, bn, act):
    a, b = _bn_affine(bn)
    return conv_scale_act(x, conv.weight, a, b, conv.stride[0], conv.padding[0], act, packed=getattr(conv, '_eg3d_packed', None))


class BasicBlock(torch.nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = torch.nn.BatchNorm2d(planes)
        self.conv2 = torch.nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = torch.nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        out = _conv_bn(x, self.conv1, self.bn1, 'relu')
        out = _conv_bn(out, self.conv2, self.bn2, 'linear')
        identity = x if self.downsample is None else _conv_bn(x, self.downsample[0], self.downsample[1], 'linear')
        return bias_act.bias_act(out + identity, None, act='relu', gain=1)      # gain=1: bias_act's default for relu is sqrt(2)


class ResNetPose(torch.nn.Module):
    """scripts/resnet/resnet.py ResNet(BasicBlock, layers, output_dims): forward(img [N,3,H,W]) -> tanh(head) [N, output_dims]."""

    def __init__(self, layers=(3, 4, 6, 3), output_dims=4):
        super().__init__()
        self.inplanes = 64
        self.conv1 = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = torch.nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.fc = torch.nn.Linear(512, 1000)
        self.fc2 = torch.nn.Linear(1000, 128)
        self.fc3 = torch.nn.Linear(128, output_dims)
        for m in self.modules():                             # resnet.py:162-167
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        self.eval()

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = torch.nn.Sequential(torch.nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), torch.nn.BatchNorm2d(planes))
        seq = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        seq += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return torch.nn.Sequential(*seq)

    def train(self, mode=True):
        """The reference only ever runs this network in eval mode (running BatchNorm statistics); training-mode statistics are not
        implemented on this path."""
        if mode:
            raise NotImplementedError('ResNetPose runs with frozen BatchNorm statistics (w_projector.py:62 .eval())')
        return super().train(False)

    def _bank_affines(self):
        """(a, b) of every eval-mode BatchNorm from four launches on the concatenated statistics (per layer that is ~10 launches on
        64...512-element tensors, forward and backward, times 36 layers); split() so that autograd joins the 36 gradients with one cat."""
        bns = [m for m in self.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        sizes = [bn.num_features for bn in bns]
        key = tuple((bn.running_mean.data_ptr(), bn.running_mean._version, bn.running_var._version) for bn in bns)
        if getattr(self, '_stat_key', None) != key:
            assert len({bn.eps for bn in bns}) == 1
            with torch.no_grad():
                self._stat_cat = (torch.cat([bn.running_mean for bn in bns]).float(), torch.cat([bn.running_var for bn in bns]).float())
            self._stat_key = key
        mean, var = self._stat_cat
        a = torch.cat([bn.weight for bn in bns]) * torch.rsqrt(var + bns[0].eps)
        b = torch.cat([bn.bias for bn in bns]) - mean * a
        for bn, ai, bi in zip(bns, torch.split(a, sizes), torch.split(b, sizes)):
            bn._eg3d_affine = (ai, bi)
        return bns

    def _pack_all(self):
        """Both packed images (forward operand, data-gradient operand) of every folded conv weight w * a[o] from ONE launch
        (eg3d_pack_conv_weights_batched with eg3d_pack_item::oscale) instead of one 11 us launch per layer: the weights are trained, so
        the images are rebuilt every step."""
        pairs = [(self.conv1, self.bn1)]
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
                if blk.downsample is not None:
                    pairs.append((blk.downsample[0], blk.downsample[1]))
        items, convs = [], []
        for conv, bn in pairs:
            w = conv.weight
            Co, Ci, kh, kw = w.shape
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and conv_scale_act_ok(Ci, w, 'relu')):
                continue
            a, _ = _bn_affine(bn)
            wf = torch.empty((Co, kh * kw * Ci), device=w.device)
            wa = torch.empty((Ci, kh * kw * Co), device=w.device)
            items.append((w.detach(), wf, wa, None, 0, a.detach().contiguous().float()))
            conv._eg3d_packed = (wf, wa)
            convs.append(conv)
        if items:
            H.pack_conv_weights_batched(items)
        return convs

    def forward(self, img):
        n, c, h, w = img.shape
        x = torch.cat([img.float(), img.new_zeros(n, 1, h, w, dtype=torch.float32)], 1).contiguous(memory_format=torch.channels_last)
        bns = self._bank_affines()
        convs = self._pack_all()
        try:
            x = _conv_bn(x, self.conv1, self.bn1, 'relu')
            x = max_pool(H.to_cl(F.pad(x, (1, 1, 1, 1), value=float('-inf'))), 3, 2)
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                x = layer(x)
        finally:
            for bn in bns:
                bn._eg3d_affine = None
            for cv in convs:
                cv._eg3d_packed = None
        x = x.mean((2, 3))
        x = F.relu(self.fc(x))
        x = F.relu(self.fc2(x))
        return torch.tanh(self.fc3(x))


def resnet34_pose(output_dims=4):
    return ResNetPose((3, 4, 6, 3), output_dims)
