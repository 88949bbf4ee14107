"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

Functional restatement of the one-shot latent encoder: Encoder4Editing(50, 'ir_se')
(models/e4e/encoders/psp_encoders.py:124-200, helpers.py:22-120, models/e4e/stylegan2/model.py:129-158) in eval mode, as a function of a
state dict with the reference's keys.  Pinned against the reference's own class by tests/golden/make_golden.py::gen_e4e (fixture
tests/golden/e4e.npz; the reference module is imported with its CUDA-extension sub-package `models.e4e.stylegan2.op` stubbed -- the
encoder never calls it); weights come from `synth_state`, never stored.
"""
import math

import torch
import torch.nn.functional as F

from .pose_net_oracle import _key_seed, _randn

STAGES = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))


def units():
    out = []
    for cin, depth, n in STAGES:
        out += [(cin, depth, 2)] + [(depth, depth, 1)] * (n - 1)
    return out


def state_shapes():
    sh = {}

    def bn(p, c):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            sh[f'{p}.{k}'] = (c,)
        sh[f'{p}.num_batches_tracked'] = ()
    sh['input_layer.0.weight'] = (64, 3, 3, 3); bn('input_layer.1', 64); sh['input_layer.2.weight'] = (64,)
    for i, (cin, d, s) in enumerate(units()):
        p = f'body.{i}'
        if cin != d:
            sh[f'{p}.shortcut_layer.0.weight'] = (d, cin, 1, 1); bn(f'{p}.shortcut_layer.1', d)
        bn(f'{p}.res_layer.0', cin)
        sh[f'{p}.res_layer.1.weight'] = (d, cin, 3, 3)
        sh[f'{p}.res_layer.2.weight'] = (d,)
        sh[f'{p}.res_layer.3.weight'] = (d, d, 3, 3)
        bn(f'{p}.res_layer.4', d)
        sh[f'{p}.res_layer.5.fc1.weight'] = (d // 16, d, 1, 1)
        sh[f'{p}.res_layer.5.fc2.weight'] = (d, d // 16, 1, 1)
    for i in range(18):
        spatial = 16 if i < 3 else (32 if i < 7 else 64)
        for j in range(int(math.log2(spatial))):
            sh[f'styles.{i}.convs.{2 * j}.weight'] = (512, 512, 3, 3)
            sh[f'styles.{i}.convs.{2 * j}.bias'] = (512,)
        sh[f'styles.{i}.linear.weight'] = (512, 512)
        sh[f'styles.{i}.linear.bias'] = (512,)
    sh['latlayer1.weight'] = (512, 256, 1, 1); sh['latlayer1.bias'] = (512,)
    sh['latlayer2.weight'] = (512, 128, 1, 1); sh['latlayer2.bias'] = (512,)
    return sh


def synth_state(seed=0, heads=(0, 1, 3, 7)):
    """Deterministic non-trivial weights (He-scaled convs, BatchNorm statistics away from identity, PReLU slopes ~0.25).  Only the style
    heads listed in `heads` get random weights (the others zeros: 18 x 5 x 9.4 MB would dominate the fixture generation for nothing)."""
    sd = {}
    for k, s in state_shapes().items():
        g = torch.Generator().manual_seed(_key_seed(k, seed))
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(0)
        elif k.endswith('running_var'):
            sd[k] = 0.5 + torch.rand(s, generator=g)
        elif k.endswith('running_mean'):
            sd[k] = 0.1 * torch.randn(s, generator=g)
        elif '.res_layer.2.' in k or k == 'input_layer.2.weight':
            sd[k] = 0.25 + 0.1 * torch.randn(s, generator=g)
        elif len(s) == 1 and k.endswith('.weight'):
            sd[k] = 1 + 0.1 * torch.randn(s, generator=g)
        elif len(s) == 1:
            sd[k] = 0.1 * torch.randn(s, generator=g)
        elif k.startswith('styles.') and int(k.split('.')[1]) not in heads:
            sd[k] = torch.zeros(s)
        elif k.endswith('linear.weight'):
            sd[k] = torch.randn(s, generator=g)
        else:
            fan_in = s[1] * (s[2] * s[3] if len(s) == 4 else 1)
            sd[k] = torch.randn(s, generator=g) * math.sqrt(2.0 / fan_in)
            if k == 'input_layer.0.weight':
                sd[k] = sd[k] / 128.0          # the projector feeds [0,255] pixels: keep the activations O(1)
            if '.res_layer.3.' in k or k.endswith('fc2.weight'):
                sd[k] = sd[k] * 0.25           # damped residual branches and gates: with plain He scaling the 24 gated residual units are
                                               # chaotic (fp32 and fp64 evaluations of the SAME weights differ by 20-50 %), useless as a parity case
    return sd


def _bn(x, sd, p):
    return F.batch_norm(x, sd[f'{p}.running_mean'], sd[f'{p}.running_var'], sd[f'{p}.weight'], sd[f'{p}.bias'], False, 0.0, 1e-5)


def trunk(sd, x):
    x = F.conv2d(x, sd['input_layer.0.weight'], padding=1)
    x = F.prelu(_bn(x, sd, 'input_layer.1'), sd['input_layer.2.weight'])
    feats = {}
    for i, (cin, d, s) in enumerate(units()):
        p = f'body.{i}'
        sc = x[:, :, ::s, ::s] if cin == d else _bn(F.conv2d(x, sd[f'{p}.shortcut_layer.0.weight'], stride=s), sd, f'{p}.shortcut_layer.1')
        y = F.conv2d(_bn(x, sd, f'{p}.res_layer.0'), sd[f'{p}.res_layer.1.weight'], padding=1)
        y = F.prelu(y, sd[f'{p}.res_layer.2.weight'])
        y = _bn(F.conv2d(y, sd[f'{p}.res_layer.3.weight'], stride=s, padding=1), sd, f'{p}.res_layer.4')
        g = y.mean((2, 3), keepdim=True)
        g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, sd[f'{p}.res_layer.5.fc1.weight'])), sd[f'{p}.res_layer.5.fc2.weight']))
        x = y * g + sc
        if i in (6, 20, 23):
            feats[i] = x
    return feats[6], feats[20], feats[23]


def style_head(sd, i, x):
    j = 0
    while f'styles.{i}.convs.{2 * j}.weight' in sd:
        x = F.leaky_relu(F.conv2d(x, sd[f'styles.{i}.convs.{2 * j}.weight'], sd[f'styles.{i}.convs.{2 * j}.bias'], stride=2, padding=1), 0.01)
        j += 1
    x = x.reshape(-1, 512)
    return F.linear(x, sd[f'styles.{i}.linear.weight'] * (1 / math.sqrt(512)), sd[f'styles.{i}.linear.bias'])


def forward(sd, x):
    """codes [N,18,512] (progressive stage = Inference)."""
    c1, c2, c3 = trunk(sd, x)
    w0 = style_head(sd, 0, c3)
    codes = [w0.clone() for _ in range(18)]
    feats = c3
    for i in range(1, 18):
        if i == 3:
            p2 = F.interpolate(c3, size=c2.shape[-2:], mode='bilinear', align_corners=True) + F.conv2d(c2, sd['latlayer1.weight'], sd['latlayer1.bias'])
            feats = p2
        elif i == 7:
            feats = F.interpolate(p2, size=c1.shape[-2:], mode='bilinear', align_corners=True) + F.conv2d(c1, sd['latlayer2.weight'], sd['latlayer2.bias'])
        codes[i] = codes[i] + style_head(sd, i, feats)
    return torch.stack(codes, 1)
