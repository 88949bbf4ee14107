"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

Pure-PyTorch CPU restatement of the three perceptual-loss networks the reference calls on the inversion path
(SURVEY.md section 8f row f1), as functions of a state dict with the original modules' keys:

  vgg16_lpips_features   the stylegan2-ada `vgg16.pt` torchscript called with return_lpips=True
                         (training/projectors/w_projector.py:50-52,112,215-219)
  vgg16_features         torchvision vgg16().features children 0..upto (training/warping_loss.py:74-105, layers='14' -> child 14)
  lpips_alex_features    lpips.LPIPS(net='alex')  (training/coaches/base_coach.py:48,111-112)

PARITY UNPINNED against the third-party packages: `lpips` (unpinned pip dependency, environment.yml:39), `torchvision` and the
NVIDIA torchscript are absent from this image and from /root/reference, and there is no network, so none of them can be run
here.  What is restated is their published algorithm:
  * VGG-16 configuration D (Simonyan & Zisserman 2015), 3x3 convs pad 1, ReLU, 2x2/2 max pooling, torchvision child numbering;
  * AlexNet features as shipped by torchvision (11x11/4 pad 2, 5x5 pad 2, 3x3 pad 1 convs; 3x3/2 max pooling), sliced after each
    ReLU as lpips.pretrained_networks.alexnet does;
  * LPIPS v0.1 (Zhang et al. 2018): ScalingLayer (x - shift) / scale on [-1,1] images, per-layer normalize_tensor
    x / (sqrt(sum_c x^2) + 1e-10), squared difference, non-negative 1x1 `lin` weights, spatial mean, sum over layers.  Returned in the
    "feature" form  f_l = sqrt(lin_l) * n_l / sqrt(H_l W_l)  (pixel-major, channel-minor), for which sum((f(a) - f(b))^2) is the
    LPIPS distance -- the form the projector consumes (w_projector.py:217-219).
The parity tests anchor on this file for the arithmetic of the HIP layers (conv + bias + ReLU epilogue, strided / large-kernel tap
lists, pooling with its arg-max tie rule, the normalisation head and their gradients).
"""
import math

import torch
import torch.nn.functional as F

VGG16_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')
LPIPS_SHIFT = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
LPIPS_SCALE = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)


def _vgg_children():
    kinds = []
    for v in VGG16_CFG:
        kinds += [('pool',)] if v == 'M' else [('conv',), ('relu',)]
    return kinds


def vgg16_run(sd, prefix, x, upto, taps=()):
    """Evaluate torchvision-numbered children 0..upto; returns (last output, {child index: output})."""
    outs = {}
    for i, kind in enumerate(_vgg_children()[:upto + 1]):
        if kind[0] == 'conv':
            x = F.conv2d(x, sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias'], padding=1)
        elif kind[0] == 'relu':
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2, 2)
        if i in taps:
            outs[i] = x
    return x, outs


def _lpips_head(taps, lins, eps=1e-10):
    feats = []
    for x, w in zip(taps, lins):
        n = x / (x.square().sum(1, keepdim=True).sqrt() + eps)
        f = n * w.reshape(1, -1, 1, 1).clamp_min(0).sqrt() / math.sqrt(x.shape[2] * x.shape[3])
        feats.append(f.permute(0, 2, 3, 1).flatten(1))
    return torch.cat(feats, 1)


def _lpips_input(img, input_range):
    x = img.float() * (2.0 / 255.0) - 1.0 if input_range == '255' else img.float()
    return (x - LPIPS_SHIFT.to(x)) / LPIPS_SCALE.to(x)


def vgg16_lpips_features(sd, img, input_range='255'):
    taps = (3, 8, 15, 22, 29)
    _, outs = vgg16_run(sd, 'net.', _lpips_input(img, input_range), taps[-1], taps)
    return _lpips_head([outs[t] for t in taps], [sd[f'lin{i}.model.1.weight'] for i in range(5)])


def vgg16_features(sd, img, upto=14):
    return vgg16_run(sd, 'features.', img.float(), upto)[0]


ALEX_SPEC = ((1, 0, 4, 2, False), (2, 3, 1, 2, True), (3, 6, 1, 1, True), (4, 8, 1, 1, False), (5, 10, 1, 1, False))


def lpips_alex_features(sd, img, input_range='pm1'):
    x = _lpips_input(img, input_range)
    taps = []
    for sl, idx, stride, pad, pool in ALEX_SPEC:
        if pool:
            x = F.max_pool2d(x, 3, 2)
        x = F.relu(F.conv2d(x, sd[f'net.slice{sl}.{idx}.weight'], sd[f'net.slice{sl}.{idx}.bias'], stride=stride, padding=pad))
        taps.append(x)
    return _lpips_head(taps, [sd[f'lin{i}.model.1.weight'] for i in range(5)])


def lpips_distance_direct(sd, a, b, trunk='alex', input_range='pm1'):
    """LPIPS computed the way the lpips package writes it (diff of normalised features -> lin -> spatial mean -> sum), as a
    cross-check of the feature form above."""
    def taps_of(img):
        x = _lpips_input(img, input_range)
        if trunk == 'vgg':
            t = (3, 8, 15, 22, 29)
            _, outs = vgg16_run(sd, 'net.', x, t[-1], t)
            return [outs[i] for i in t]
        res = []
        for sl, idx, stride, pad, pool in ALEX_SPEC:
            if pool:
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(F.conv2d(x, sd[f'net.slice{sl}.{idx}.weight'], sd[f'net.slice{sl}.{idx}.bias'], stride=stride, padding=pad))
            res.append(x)
        return res
    total = 0
    for i, (xa, xb) in enumerate(zip(taps_of(a), taps_of(b))):
        na = xa / (xa.square().sum(1, keepdim=True).sqrt() + 1e-10)
        nb = xb / (xb.square().sum(1, keepdim=True).sqrt() + 1e-10)
        d = F.conv2d((na - nb).square(), sd[f'lin{i}.model.1.weight'].clamp_min(0))
        total = total + d.mean((2, 3))
    return total.reshape(-1)
