"""CPU checks of the perceptual-loss network restatement (oracle/loss_nets_oracle.py) and of the host-side tap lists of
inv3d_amd/loss_nets.py.  The third-party packages are absent (parity unpinned, see the oracle header), so what can be pinned on
CPU is self-consistency: the feature form the projector consumes against LPIPS written the way the lpips package writes it,
and the tap lists of the strided / large-kernel convolutions against torch.nn.functional.conv2d."""
import torch
import torch.nn.functional as F

from oracle import loss_nets_oracle as LO


def test_feature_form_equals_direct_lpips():
    from inv3d_amd import loss_nets as LN
    g = torch.Generator().manual_seed(0)
    for trunk, cls, fn, size in (('alex', LN.LPIPSAlex, LO.lpips_alex_features, 96), ('vgg', LN.VGG16LPIPS, LO.vgg16_lpips_features, 32)):
        sd = cls(input_range='pm1').state_dict()
        a = torch.rand(2, 3, size, size, generator=g) * 2 - 1
        b = (a + 0.3 * torch.randn(a.shape, generator=g)).clamp(-1, 1)
        d_feat = (fn(sd, a, input_range='pm1') - fn(sd, b, input_range='pm1')).square().sum(1)
        d_dir = LO.lpips_distance_direct(sd, a, b, trunk)
        assert torch.allclose(d_feat, d_dir, rtol=1e-4, atol=1e-7), (trunk, d_feat, d_dir)
        assert (d_feat > 0).all()


def test_state_dict_keys_follow_the_original_modules():
    from inv3d_amd import loss_nets as LN
    v = LN.VGG16Features().state_dict()
    assert [k for k in v if k.endswith('weight')] == [f'features.{i}.weight' for i in (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)]
    a = LN.LPIPSAlex().state_dict()
    for k in ('net.slice1.0.weight', 'net.slice2.3.weight', 'net.slice3.6.weight', 'net.slice4.8.weight', 'net.slice5.10.weight', 'lin4.model.1.weight'):
        assert k in a
    assert a['net.slice1.0.weight'].shape == (64, 3, 11, 11) and a['lin1.model.1.weight'].shape == (1, 192, 1, 1)


def _emulate(x, w, classes, in_stride, out_stride, Ho, Wo, transpose_roles):
    """What the implicit-GEMM kernel computes from a class list (include/eg3d_hip.h): acc[ay,ax,o] = sum_t x[ay*is+dy, ax*is+dx, :] . w[o, wtap, :]
    written (accumulated) at (ay*os+py, ax*os+px)."""
    N, C, H, W = x.shape
    wt = w if not transpose_roles else w.transpose(0, 1)            # [O', I', kh, kw]
    O = wt.shape[0]
    kw_ = w.shape[3]
    out = torch.zeros(N, O, Ho, Wo)
    for c in classes:
        for ay in range(c.Ha):
            for ax in range(c.Wa):
                acc = torch.zeros(N, O)
                for t in range(c.ntaps):
                    y, xx = ay * in_stride + c.dy[t], ax * in_stride + c.dx[t]
                    if 0 <= y < H and 0 <= xx < W:
                        acc += x[:, :, y, xx] @ wt[:, :, c.wtap[t] // kw_, c.wtap[t] % kw_].T
                oy, ox = ay * out_stride + c.out_py, ax * out_stride + c.out_px
                if oy < Ho and ox < Wo:
                    out[:, :, oy, ox] += acc
    return out


def test_strided_tap_lists_match_conv2d_and_its_adjoint():
    from inv3d_amd import loss_nets as LN
    g = torch.Generator().manual_seed(1)
    for (k, s, p, h, w) in ((11, 4, 2, 23, 19), (5, 1, 2, 6, 7), (3, 2, 1, 7, 8), (3, 1, 1, 5, 5)):
        x = torch.randn(1, 2, h, w, generator=g, dtype=torch.float64).requires_grad_(True)
        wt = torch.randn(3, 2, k, k, generator=g, dtype=torch.float64)
        y = F.conv2d(x, wt, stride=s, padding=p)
        Ho, Wo = y.shape[2:]
        cls = LN._classes_strided(Ho, Wo, k, k, p)
        assert all(c.ntaps <= 9 for c in cls)
        ye = _emulate(x.detach().float(), wt.float(), cls, s, 1, Ho, Wo, False)
        assert torch.allclose(ye.double(), y.detach(), atol=1e-4), (k, s)
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(dy)
        acls, _ = LN._classes_strided_adjoint(h, w, k, k, s, p)
        dxe = _emulate(dy.float(), wt.float(), acls, 1, s, h, w, True)
        assert torch.allclose(dxe.double(), x.grad, atol=1e-4), (k, s, 'adjoint')
