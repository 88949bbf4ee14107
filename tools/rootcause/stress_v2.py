#!/usr/bin/env python3
"""Launch-to-launch bit-identity stress of every LDS-DMA / raw-barrier convolution kernel at full-size layer shapes (round-6 root-cause work).

Every case launches one kernel instantiation LAUNCHES times on the same operands and compares the (atomic-free) output tensor of every launch with
the first one, on the device.  A kernel whose LDS protocol is sound gives 0 differing launches; a write-after-read race on the weight ring (an LDS read
still in flight when the next LDS-DMA lands in its slot) shows up as a handful of launches that differ in a few elements.

  EG3D_LIBNAME=libeg3d_hip_xyz.so python tools/rootcause/stress_v2.py [--launches 5000] [--cases v2_fwd8,...] [--json out.json]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, '3dgan-inversion_amd'))
import torch                                                    # noqa: E402
from inv3d_amd import hipops as H, _lib as L                    # noqa: E402

DEV = 'cuda'


def operands(n, ci, h, w, co, seed=1, adj=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, ci, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(n, ci, generator=g)).to(DEV)
    d = (0.5 + torch.rand(n, co, generator=g)).to(DEV)
    noise, strength = torch.randn(h, w, generator=g).to(DEV), torch.tensor(0.3, device=DEV)
    bias = (0.1 * torch.randn(co, generator=g)).to(DEV)
    aimg = H.split_activation(x, H.absmax(x), in_scale=s)
    wimg = H.split_weight(H.pack_weight_fwd(wt), co, ci, 9)
    return dict(x=x, wt=wt, s=s, d=d, noise=noise, strength=strength, bias=bias, aimg=aimg, wimg=wimg, g=g)


def case_v2(n, ci, h, co, rows, kind, khalves=True, rgb=False, products=3):
    """conv_v2_kernel<9, products == 3, false, rows / 2, rgb, KH>: forward epilogue | data-gradient epilogue | + the producer's activation backward."""
    o = operands(n, ci, h, h, co)
    cls = H.classes_corr(h, h, 3, 3, 1)
    H.V2_KHALVES = khalves
    out = H.empty_cl(n, co, h, h, DEV)
    amax = torch.zeros(1, device=DEV)
    if kind == 'fwd':
        y4 = H.empty_cl(n, 4, h, h, DEV) if rgb else None
        w4 = (torch.randn(4, co, generator=o['g']) / math.sqrt(co)).to(DEV) if rgb else None
        s4 = (1 + 0.5 * torch.randn(n, co, generator=o['g'])).to(DEV) if rgb else None
        b4 = torch.zeros(4, device=DEV) if rgb else None
        kw = dict(epi=L.EPI_FWD, out_scale=o['d'], bias=o['bias'], noise=o['noise'], noise_nstride=0, noise_strength=o['strength'], act='lrelu', alpha=0.2, gain=1.4,
                  clamp=-1.0, out_amax=amax, patch_rows=rows, products=products)
        if rgb:
            kw['rgb_head'] = (w4, s4, b4, y4, -1.0, 3)

        def launch():
            H.conv_v2(o['aimg'], o['wimg'], out, cls, **kw)
            return [out] + ([y4] if rgb else [])
    else:
        xin = torch.randn(n, co, h, h, generator=torch.Generator().manual_seed(6)).to(DEV).contiguous(memory_format=torch.channels_last)
        ds = torch.zeros(n, co, device=DEV)
        ab = None
        if kind == 'bwd_act':
            ab = H.ActBwdSpec(d=o['d'], bias=o['bias'], noise=o['noise'], noise_nstride=0, noise_strength=o['strength'], act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0,
                              dbias=torch.zeros(co, device=DEV), dd=torch.zeros(n, co, device=DEV), dnoise=torch.zeros(h, h, device=DEV), dnoise_nstride=0,
                              dstrength=torch.zeros(1, device=DEV))

        def launch():
            H.conv_v2(o['aimg'], o['wimg'], out, cls, epi=L.EPI_BWD, out_scale=o['d'], xin=xin, ds=ds, out_amax=amax, patch_rows=rows, act_bwd=ab, products=products)
            return [out]
    return launch


def case_v2_convT(n, ci, h, co):
    """conv_v2_kernel<4 | 2 | 1>: stride-2 transposed conv as four parity classes, plain-store epilogue."""
    o = operands(n, ci, h, h, co)
    cls, hz, wz = H.classes_convT(h, h, 3, 3, 2)
    z = H.empty_cl(n, co, hz, wz, DEV)

    def launch():
        H.conv_v2(o['aimg'], o['wimg'], z, cls, out_stride=2, epi=L.EPI_STORE, patch_rows=8)
        return [z]
    return launch


def case_up2(n, ci, h, co, rows=8):
    o = operands(n, ci, h, h, co)
    z = H.empty_cl(n, co, 2 * h + 1, 2 * h + 1, DEV)
    z.zero_()

    def launch():
        H.conv_up2(o['aimg'], o['wimg'], z, Hc=h, Wc=h, epi=L.EPI_STORE, patch_rows=rows)
        return [z]
    return launch


def case_s2adj(n, ci, h, co, v3=False):
    """layer ci -> co, input h x h, output 2h x 2h: the data gradient dx [n, ci, h, h]."""
    g_ = torch.Generator().manual_seed(51)
    dz = (torch.randn(n, co, 2 * h, 2 * h, generator=g_) * 1e-3).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, generator=g_) / math.sqrt(ci * 9)).to(DEV)
    s = (1 + 0.5 * torch.randn(n, ci, generator=g_)).to(DEV)
    xin = torch.randn(n, ci, h, h, generator=g_).to(DEV).contiguous(memory_format=torch.channels_last)
    gimg = H.fir44_adjoint_split(dz, H.absmax(dz), gain=4.0)
    wimg = H.split_weight(H.pack_weight_adj(wt), ci, co, 9)
    dx, ds = H.empty_cl(n, ci, h, h, DEV), torch.zeros(n, ci, device=DEV)
    cls = H.classes_convT_adjoint(h, h, 3, 3, 2)

    def launch():
        H.conv_v2_s2adj(gimg, wimg, dx, cls, epi=L.EPI_BWD, out_scale=s, xin=xin, ds=ds, **({'v3': True} if v3 else {}))
        return [dx]
    return launch


def case_v3(n, ci, h, co, plan):
    o = operands(n, ci, h, h, co)
    cls = H.classes_corr(h, h, 3, 3, 1)
    out = H.empty_cl(n, co, h, h, DEV)
    amax = torch.zeros(1, device=DEV)

    def launch():
        H.conv_v3(o['aimg'], o['wimg'], out, cls, plan=plan, out_amax=amax, epi=L.EPI_FWD, out_scale=o['d'], bias=o['bias'], noise=o['noise'], noise_nstride=0,
                  noise_strength=o['strength'], act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0)
        return [out]
    return launch


def case_wgrad_v2(n, ci, co, h):
    g_ = torch.Generator().manual_seed(7)
    x = torch.randn(n, ci, h, h, generator=g_).to(DEV).contiguous(memory_format=torch.channels_last)
    s = (torch.rand(n, ci, generator=g_) + 0.5).to(DEV)
    dy = (torch.randn(n, co, h, h, generator=g_) * 1e-6).to(DEV).contiguous(memory_format=torch.channels_last)
    ximg = H.split_activation(x, H.absmax(x), in_scale=s)
    gimg = H.split_activation(dy, H.absmax(dy))
    cls = H.classes_corr(h, h, 3, 3, 1)

    def launch():
        slabs = H.conv_wgrad_v2_slabs(gimg, ximg, cls, products=3, row_groups=0)
        return [slabs if torch.is_tensor(slabs) else slabs[0]]
    return launch


CASES = {
    # the dominant kernel: SR block 1 conv1 forward (512^2 x 128 -> 128), with and without the 1x1 head, and its data gradients
    'v2_fwd8_512x128': lambda: case_v2(1, 128, 512, 128, 8, 'fwd'),
    'v2_fwd8_rgb_512x128': lambda: case_v2(1, 128, 512, 128, 8, 'fwd', rgb=True),
    'v2_bwd8_512x128': lambda: case_v2(1, 128, 512, 128, 8, 'bwd'),
    'v2_bwdact8_512x128': lambda: case_v2(1, 128, 512, 128, 8, 'bwd_act'),
    'v2_fwd8_256x256': lambda: case_v2(1, 256, 256, 256, 8, 'fwd'),
    'v2_f16x1_fwd8_512x128': lambda: case_v2(1, 128, 512, 128, 8, 'fwd', products=1),
    # 4-row patches: four-wave form and the eight-wave K-halves form
    'v2_fwd4_kh1_256x128': lambda: case_v2(1, 128, 256, 128, 4, 'fwd', khalves=False),
    'v2_fwd4_kh1_128x256': lambda: case_v2(1, 256, 128, 256, 4, 'fwd', khalves=False),
    'v2_bwd4_kh1_128x256': lambda: case_v2(1, 256, 128, 256, 4, 'bwd', khalves=False),
    'v2_bwdact4_kh1_256x128': lambda: case_v2(1, 128, 256, 128, 4, 'bwd_act', khalves=False),
    'v2_fwd4_kh2_256x128': lambda: case_v2(1, 128, 256, 128, 4, 'fwd', khalves=True),
    'v2_fwd4_kh2_128x256': lambda: case_v2(1, 256, 128, 256, 4, 'fwd', khalves=True),
    'v2_bwd4_kh2_128x256': lambda: case_v2(1, 256, 128, 256, 4, 'bwd', khalves=True),
    # 2-row patches
    'v2_fwd2_64x512': lambda: case_v2(1, 512, 64, 512, 2, 'fwd'),
    'v2_bwd2_64x512': lambda: case_v2(1, 512, 64, 512, 2, 'bwd'),
    # tap classes 4 / 2 / 2 / 1
    'v2_convT_128x256': lambda: case_v2_convT(1, 256, 128, 128),
    # the up-sampling layers: fused-parity forward, parity-split adjoint
    'up2_256x256to128': lambda: case_up2(1, 256, 256, 128),
    'up2_128x256to256': lambda: case_up2(1, 256, 128, 256),
    'up2r4_256x256to128': lambda: case_up2(1, 256, 256, 128, rows=4),
    'up2r4_128x256to128': lambda: case_up2(1, 256, 128, 128, rows=4),
    'up2r4_128x32to256': lambda: case_up2(1, 32, 128, 256, rows=4),
    's2adj_256x256from128': lambda: case_s2adj(1, 256, 256, 128),
    's2adj_128x256from256': lambda: case_s2adj(1, 256, 128, 256),
    's2adj_v3_64x512': lambda: case_s2adj(1, 512, 64, 256, v3=True),
    # wave-split kernel
    'v3_64x512_r4w4': lambda: case_v3(1, 512, 64, 512, (4, 4)),
    'v3_64x512_r2w8': lambda: case_v3(1, 512, 64, 512, (2, 8)),
    'v3_32x512_r2w8': lambda: case_v3(1, 512, 32, 512, (2, 8)),
    # weight gradient
    'wgrad_v2_256x128': lambda: case_wgrad_v2(1, 128, 128, 256),
    'wgrad_v2_128x256': lambda: case_wgrad_v2(1, 256, 256, 128),
}


def run_case(name, launches):
    launch = CASES[name]()
    refs = [t.clone() for t in launch()]
    torch.cuda.synchronize()
    nbad = torch.zeros(1, device=DEV, dtype=torch.int64)           # launches that differ
    nelem = torch.zeros(1, device=DEV, dtype=torch.int64)          # differing elements, all launches
    maxrel = torch.zeros(1, device=DEV)
    scale = float(refs[0].abs().max())
    t0 = time.time()
    for _ in range(launches):
        outs = launch()
        for o, r in zip(outs, refs):
            ne = (o != r)
            c = ne.sum()
            nelem += c
            nbad += (c > 0).to(torch.int64)
            maxrel = torch.maximum(maxrel, ((o - r).abs().max() / scale).reshape(1))
    torch.cuda.synchronize()
    return dict(case=name, launches=launches, differing_launches=int(nbad), differing_elements=int(nelem), max_rel_diff=float(maxrel), seconds=round(time.time() - t0, 1))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=5000)
    ap.add_argument('--cases', default='')
    ap.add_argument('--json', default='')
    a = ap.parse_args()
    names = [c for c in a.cases.split(',') if c] or list(CASES)
    res = []
    for nm in names:
        try:
            r = run_case(nm, a.launches)
        except Exception as e:                                  # noqa: BLE001
            r = dict(case=nm, error=repr(e)[:300])
        r['lib'] = os.path.basename(L.LIB_PATH)
        print(json.dumps(r), flush=True)
        res.append(r)
        torch.cuda.empty_cache()
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(res, f, indent=1)
    bad = [r for r in res if r.get('differing_launches', 0) or 'error' in r]
    print('SUMMARY lib=%s cases=%d not-clean=%d' % (os.path.basename(L.LIB_PATH), len(res), len(bad)))
