"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

Functional restatement of the in-loop pose estimator: ResNet-34 + 512->1000->128->D head of
scripts/resnet/resnet.py:124-230 in eval mode (BatchNorm on running statistics, training/projectors/w_projector.py:62), as a
function of a state dict with the reference's keys.  Pinned against the reference's own class by
tests/golden/make_golden.py::gen_pose_net (fixture tests/golden/pose_net.npz: input, output, gradients of a few parameters);
weights come from `synth_state` below, never stored.
"""
import math

import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)


def _key_seed(name, seed):
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) % (1 << 64)
    return (h ^ (seed * 0x9E3779B97F4A7C15)) % (1 << 63)


def _randn(name, seed, shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(_key_seed(name, seed)))


def state_shapes(output_dims=4):
    sh = {'conv1.weight': (64, 3, 7, 7)}

    def bn(prefix, c):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            sh[f'{prefix}.{k}'] = (c,)
        sh[f'{prefix}.num_batches_tracked'] = ()
    bn('bn1', 64)
    inpl = 64
    for li, (nb, pl) in enumerate(zip(LAYERS, PLANES), 1):
        for b in range(nb):
            p = f'layer{li}.{b}'
            stride = 2 if (b == 0 and li > 1) else 1
            sh[f'{p}.conv1.weight'] = (pl, inpl, 3, 3); bn(f'{p}.bn1', pl)
            sh[f'{p}.conv2.weight'] = (pl, pl, 3, 3); bn(f'{p}.bn2', pl)
            if stride != 1 or inpl != pl:
                sh[f'{p}.downsample.0.weight'] = (pl, inpl, 1, 1); bn(f'{p}.downsample.1', pl)
            inpl = pl
    sh.update({'fc.weight': (1000, 512), 'fc.bias': (1000,), 'fc2.weight': (128, 1000), 'fc2.bias': (128,),
               'fc3.weight': (output_dims, 128), 'fc3.bias': (output_dims,)})
    return sh


def synth_state(seed=0, output_dims=4):
    """Deterministic non-trivial weights: He-scaled convolutions / linears, BatchNorm gamma 1 +- 0.1, beta, mean +- 0.1, var in [0.5, 1.5]."""
    sd = {}
    for k, s in state_shapes(output_dims).items():
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(0)
        elif k.endswith('running_var'):
            sd[k] = 0.5 + torch.rand(s, generator=torch.Generator().manual_seed(_key_seed(k, seed)))
        elif len(s) == 1 and ('bn' in k or 'downsample.1' in k):
            sd[k] = (1.0 if k.endswith('.weight') else 0.0) + 0.1 * _randn(k, seed, s)
        elif len(s) == 1:
            sd[k] = 0.05 * _randn(k, seed, s)
        else:
            fan_in = math.prod(s[1:])
            sd[k] = _randn(k, seed, s) * math.sqrt(2.0 / fan_in) * (0.003 if k == 'fc3.weight' else 1.0)    # keeps the tanh output unsaturated
    return sd


def _bn(sd, p, x):
    return F.batch_norm(x, sd[f'{p}.running_mean'], sd[f'{p}.running_var'], sd[f'{p}.weight'], sd[f'{p}.bias'], False, 0.0, 1e-5)


def forward(sd, img):
    """resnet.py:205-227 (_forward_impl) with BasicBlock.forward (:57-72)."""
    x = F.relu(_bn(sd, 'bn1', F.conv2d(img.float(), sd['conv1.weight'], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nb in enumerate(LAYERS, 1):
        for b in range(nb):
            p = f'layer{li}.{b}'
            stride = 2 if (b == 0 and li > 1) else 1
            out = F.relu(_bn(sd, f'{p}.bn1', F.conv2d(x, sd[f'{p}.conv1.weight'], stride=stride, padding=1)))
            out = _bn(sd, f'{p}.bn2', F.conv2d(out, sd[f'{p}.conv2.weight'], padding=1))
            idn = x
            if f'{p}.downsample.0.weight' in sd:
                idn = _bn(sd, f'{p}.downsample.1', F.conv2d(x, sd[f'{p}.downsample.0.weight'], stride=stride))
            x = F.relu(out + idn)
    x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    x = F.relu(F.linear(x, sd['fc.weight'], sd['fc.bias']))
    x = F.relu(F.linear(x, sd['fc2.weight'], sd['fc2.bias']))
    return torch.tanh(F.linear(x, sd['fc3.weight'], sd['fc3.bias']))
