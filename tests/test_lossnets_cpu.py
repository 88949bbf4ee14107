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


def test_direct_conv_weight_packing_matches_conv2d_and_its_adjoint():
    """Host side of eg3d_conv3x3_direct (inv3d_amd/loss_nets.py::_pack_direct, _direct_group): the packed image [Co/G][Ci/4][9][4][G], read the way
    the kernel reads it (tap = ky * 3 + kx, correlation, zero padding 1), reproduces F.conv2d; the image of the flipped, channel-transposed weights
    reproduces the data gradient -- for every group width the launcher can pick, input channels padded to a multiple of 4."""
    from inv3d_amd import loss_nets as LN
    g = torch.Generator().manual_seed(3)

    def run_packed(x, wp, co, G):            # x [N,Cip,H,W]; the kernel's arithmetic, one tap at a time
        n, cip, h, w = x.shape
        xp = F.pad(x, (1, 1, 1, 1))
        out = torch.zeros(n, co, h, w, dtype=x.dtype)
        for cog in range(co // G):
            for cq in range(cip // 4):
                for tap in range(9):
                    ky, kx = divmod(tap, 3)
                    win = xp[:, 4 * cq:4 * cq + 4, ky:ky + h, kx:kx + w]                       # [N,4,H,W]
                    wv = wp[cog, cq, tap]                                                     # [4][G]
                    out[:, cog * G:(cog + 1) * G] += torch.einsum('njhw,jg->nghw', win, wv)
        return out

    for (ci, co, hw) in ((4, 16, 8), (16, 32, 6), (3, 8, 4)):
        w = torch.randn(co, ci, 3, 3, generator=g).double()
        cip = (ci + 3) // 4 * 4
        x = torch.randn(2, ci, hw, hw, generator=g).double()
        xpad = torch.cat([x, x.new_zeros(2, cip - ci, hw, hw)], 1)
        ref = F.conv2d(x, w, padding=1)
        for G in (1, 2, 4):
            wp = LN._pack_direct(w, G).double()
            assert tuple(wp.shape) == (co // G, cip // 4, 9, 4, G)
            assert torch.allclose(run_packed(xpad, wp, co, G), ref, atol=1e-12)
        if ci % 4 == 0:                      # data gradient: dx = conv(dy, flip(W)^T)
            dy = torch.randn(2, co, hw, hw, generator=g).double()
            xr = x.clone().requires_grad_(True)
            (F.conv2d(xr, w, padding=1) * dy).sum().backward()
            for G in (1, 2, 4):
                wa = LN._pack_direct(w.flip(2, 3).permute(1, 0, 2, 3), G).double()
                assert torch.allclose(run_packed(dy, wa, ci, G), xr.grad, atol=1e-12)
    # group width: widest that leaves >= 32 k threads; the contraction is dealt to four waves from 16 input channels on
    assert LN._direct_group(16, 128 * 128, 4) == 4 and LN._direct_group(64, 32 * 32, 32) == 4 and LN._direct_group(32, 32 * 32, 4) == 1
