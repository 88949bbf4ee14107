"""impl='ref' of the three L1 operators: plain-torch composites owned by the product (any device, differentiable to any order through
autograd), selected ONLY by an explicit `impl='ref'` argument -- the keyword the reference's own operators carry (torch_utils/ops/bias_act.py:84-88,
upfirdn2d.py:160-164, filtered_lrelu.py:113-120).  There is NO automatic fallback: impl='cuda' (the default) raises on CPU tensors and when the HIP
library is missing.  These composites follow the operators' mathematical definitions; they do not import oracle/ (test infrastructure) and the
fused generator path never calls them."""
import torch
import torch.nn.functional as F

_ACT = {
    'linear': lambda x, a: x,
    'relu': lambda x, a: F.relu(x),
    'lrelu': lambda x, a: F.leaky_relu(x, a),
    'tanh': lambda x, a: torch.tanh(x),
    'sigmoid': lambda x, a: torch.sigmoid(x),
    'elu': lambda x, a: F.elu(x),
    'selu': lambda x, a: F.selu(x),
    'softplus': lambda x, a: F.softplus(x),
    'swish': lambda x, a: torch.sigmoid(x) * x,
}


def bias_act_ref(x, b, dim, act, alpha, gain, clamp):
    """clamp(act(x + b) * gain): b broadcast along `dim`, clamp < 0 = none."""
    if b is not None:
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.dim())])
    x = _ACT[act](x, alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def upfirdn2d_ref(x, f, up, down, padding, flip_filter, gain):
    """Zero-insert by up = (ux, uy), pad / crop by padding = (px0, px1, py0, py1), correlate with the flipped filter (a true convolution unless
    flip_filter), keep every down = (dx, dy)-th sample.  f: None (identity), 1-D separable [taps] or 2-D [fh, fw]."""
    n, c, h, w = x.shape
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, padding
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    x = x.reshape(n, c, h, 1, w, 1)
    x = F.pad(x, [0, ux - 1, 0, 0, 0, uy - 1]).reshape(n, c, h * uy, w * ux)
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0):x.shape[2] - max(-py1, 0), max(-px0, 0):x.shape[3] - max(-px1, 0)]
    f = (f * (gain ** (f.dim() / 2))).to(x.dtype)
    if not flip_filter:
        f = f.flip(list(range(f.dim())))
    if f.dim() == 2:
        x = F.conv2d(x, f[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        x = F.conv2d(x, f[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        x = F.conv2d(x, f[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return x[:, :, ::dy, ::dx]


def filtered_lrelu_ref(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter):
    """bias -> up-sample (fu, gain up^2) -> leaky ReLU * gain, clamp -> down-sample (fd)."""
    px0, px1, py0, py1 = padding
    x = bias_act_ref(x, b, 1, 'linear', 0.0, 1.0, -1.0)
    x = upfirdn2d_ref(x, fu, (up, up), (1, 1), (px0, px1, py0, py1), flip_filter, float(up) ** 2)
    x = bias_act_ref(x, None, 1, 'lrelu', slope, gain, -1.0 if clamp is None else clamp)
    return upfirdn2d_ref(x, fd, (1, 1), (down, down), (0, 0, 0, 0), flip_filter, 1.0)
