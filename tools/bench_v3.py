"""Stand-alone timing of the wave-split pre-split conv (csrc/conv_v3.hip) against what it replaces, on the backbone's under-filled 3x3 layers
at one image per GPU, forward with the fused epilogue, COLD weights (each launch of a replayed graph reads another copy of the weights: inside
the step every layer's weight image arrives from HBM / MALL).   python tools/bench_v3.py [n_images]"""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import hipops as H, _lib as L, fused as F
dev = torch.device('cuda')
NWT = 8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(0)


def timed(fn, reps=5):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for k in range(NWT):
            fn(k)
        st.synchronize()
        gr = torch.cuda.CUDAGraph()
        with H.capture_guard(), torch.cuda.graph(gr, stream=st):
            for k in range(NWT):
                fn(k)
        gr.replay(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record(st); gr.replay(); e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / NWT)
    return best


for (ci, res, co) in ((512, 32, 512), (512, 64, 512), (256, 128, 256), (128, 256, 128)):
    ws = [(torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(dev) for _ in range(NWT)]
    wfs = [H.pack_weight_fwd(w) for w in ws]
    wimgs = [H.split_weight(wf, co, ci, 9) for wf in wfs]
    wps = [H.split_weight_pieces(wf) for wf in wfs]
    s = (1 + 0.5 * torch.randn(N, ci, generator=g)).to(dev)
    d = (0.5 + torch.rand(N, co, generator=g)).to(dev)
    bias = torch.zeros(co, device=dev)
    strength = torch.tensor(0.1, device=dev)
    x = torch.randn(N, ci, res, res, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ax = H.absmax(x)
    noise = torch.randn(res, res, generator=g).to(dev)
    cls = H.classes_corr(res, res, 3, 3, 1)
    out = H.empty_cl(N, co, res, res, dev)
    amax = torch.zeros(1, device=dev)
    epi = dict(noise=noise, noise_nstride=0, noise_strength=strength, bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0)
    gf = 2.0 * N * res * res * ci * co * 9 / 1e9
    ks_old = F._auto_ksplit(cls, N, co, ci)
    z = torch.zeros(N, co, res, res, device=dev).contiguous(memory_format=torch.channels_last)
    aimg = H.split_activation(x, ax, in_scale=s)

    def old(k):
        if ks_old > 1:
            z.zero_()
            H.conv_atomic(x, wfs[k], ci, co, z, cls, in_scale=s, ksplit=ks_old, precision='f16x3', w_pieces=wps[k])
            H.epilogue_fwd(z, out, d=d, out_amax=amax, **epi)
        else:
            H.conv_igemm(x, wfs[k], ci, co, out, cls, in_scale=s, epi=L.EPI_FWD, out_scale=d, precision='f16x3', out_amax=amax, w_pieces=wps[k], **epi)
    ref = None
    t = timed(old)
    old(0); ref = out.clone()
    print(f'{res}^2 x {ci}->{co} N={N} ({gf:.1f} GF): igemm ks {ks_old} (+fill+finish) {t:6.1f} us ({gf / t * 1e3:4.0f} TF/s)', flush=True)
    t = timed(lambda k: H.split_activation(x, ax, in_scale=s))
    print(f'      operand split pass {t:5.1f} us', flush=True)
    for rows in (8, 4, 2):
        if co % 128:
            continue
        def v2(k):
            H.conv_v2(aimg, wimgs[k], out, cls, epi=L.EPI_FWD, out_scale=d, out_amax=amax, patch_rows=rows, **epi)
        t = timed(v2)
        v2(0)
        err = float((out - ref).abs().max() / ref.abs().max())
        print(f'      conv_v2 rows {rows}: {t:6.1f} us ({gf / t * 1e3:4.0f} TF/s)  diff {err:.1e}', flush=True)
    for plan in ((4, 4), (2, 4), (2, 8)):
        for products in (3, 1):
            def v3(k):
                H.conv_v3(aimg, wimgs[k], out, cls, plan=plan, epi=L.EPI_FWD, out_scale=d, out_amax=amax, products=products, **epi)
            t = timed(v3)
            v3(0)
            err = float((out - ref).abs().max() / ref.abs().max())
            print(f'      conv_v3 rows {plan[0]} waves {plan[1]} products {products}: {t:6.1f} us ({gf / t * 1e3:4.0f} TF/s, executed {gf * products / t * 1e3:4.0f})  diff {err:.1e}',
                  flush=True)



# ---- the up-sampling layers' data gradient (stride-2 adjoint on parity-split images): wave-split kernel vs the loader-split kernel ----------
print('--- stride-2 adjoint (data gradient of the up layers, EPI_BWD) ---', flush=True)
for (ci, res, co) in ((512, 32, 512), (512, 64, 256), (256, 128, 128)):
    ws = [(torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(dev) for _ in range(NWT)]
    was = [H.pack_weight_adj(w) for w in ws]
    wimgs = [H.split_weight(wa, ci, co, 9) for wa in was]
    wps = [H.split_weight_pieces(wa) for wa in was]
    s = (1 + 0.5 * torch.randn(N, ci, generator=g)).to(dev)
    xin = torch.randn(N, ci, res, res, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dz = (torch.randn(N, co, 2 * res, 2 * res, generator=g) * 1e-3).to(dev).contiguous(memory_format=torch.channels_last)
    amax = H.absmax(dz)
    cls = H.classes_convT_adjoint(res, res, 3, 3, 2)
    dx, ds = H.empty_cl(N, ci, res, res, dev), torch.zeros(N, ci, device=dev)
    gf = 2.0 * N * res * res * ci * co * 9 / 1e9
    gfull = H.empty_cl(N, co, 2 * res + 1, 2 * res + 1, dev)
    ks_old = F._auto_ksplit(cls, N, ci, co)
    z = torch.zeros(N, ci, res, res, device=dev).contiguous(memory_format=torch.channels_last)

    def old(k):
        H.upconv_epilogue_fwd(dz, gfull, pad0=2, fir_gain=4.0)
        if ks_old > 1:
            z.zero_()
            H.conv_atomic(gfull, was[k], co, ci, z, cls, in_stride=2, ksplit=ks_old, precision='f16x3', a_amax=amax, w_pieces=wps[k], a_amax_mul=4.0)
            H.dgrad_finish(z, xin, s, dx, ds=ds)
        else:
            H.conv_igemm(gfull, was[k], co, ci, dx, cls, in_stride=2, epi=L.EPI_BWD, out_scale=s, xin=xin, ds=ds, precision='f16x3', a_amax=amax, a_amax_mul=4.0,
                         w_pieces=wps[k])
    t = timed(old)
    old(0); ref = dx.clone()
    print(f'{2 * res}^2 -> {res}^2 x {co}->{ci} N={N} ({gf:.1f} GF): FIR adjoint + igemm ks {ks_old} (+finish) {t:6.1f} us ({gf / t * 1e3:4.0f} TF/s)', flush=True)
    t = timed(lambda k: H.fir44_adjoint_split(dz, amax, gain=4.0))
    print(f'      fir44_adjoint_split {t:5.1f} us', flush=True)
    gimg = H.fir44_adjoint_split(dz, amax, gain=4.0)
    for products in (3, 1):
        def v3(k):
            H.conv_v2_s2adj(gimg, wimgs[k], dx, cls, epi=L.EPI_BWD, out_scale=s, xin=xin, ds=ds, products=products, v3=True)
        t = timed(v3)
        v3(0)
        err = float((dx - ref).abs().max() / ref.abs().max())
        print(f'      conv_v3_s2adj products {products}: {t:6.1f} us ({gf / t * 1e3:4.0f} TF/s, executed {gf * products / t * 1e3:4.0f})  diff {err:.1e}', flush=True)


# ---- the 4^2 .. 16^2 layers (split-K, atomic accumulation into a zeroed buffer): weight-streaming kernel vs the loader-split kernel ----------
print('--- low-resolution 3x3 layers, EPI_ATOMIC (zero fill not included) ---', flush=True)
for (ci, res, co) in ((512, 4, 512), (512, 8, 512), (512, 16, 512), (512, 32, 512)):
    ws_ = [(torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(dev) for _ in range(NWT)]
    wfs = [H.pack_weight_fwd(w) for w in ws_]
    wimgs = [H.split_weight(wf, co, ci, 9) for wf in wfs]
    wps = [H.split_weight_pieces(wf) for wf in wfs]
    s = (1 + 0.5 * torch.randn(N, ci, generator=g)).to(dev)
    x = torch.randn(N, ci, res, res, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ax = H.absmax(x)
    cls = H.classes_corr(res, res, 3, 3, 1)
    z = torch.zeros(N, co, res, res, device=dev).contiguous(memory_format=torch.channels_last)
    gf = 2.0 * N * res * res * ci * co * 9 / 1e9
    ks_old = F._auto_ksplit(cls, N, co, ci)

    def old(k):
        H.conv_atomic(x, wfs[k], ci, co, z, cls, in_scale=s, ksplit=ks_old, precision='f16x3', w_pieces=wps[k])
    t = timed(old)
    z.zero_(); old(0); ref = z.clone()
    print(f'{res}^2 x {ci}->{co} N={N} ({gf:.2f} GF): igemm ks {ks_old} {t:6.1f} us', flush=True)
    for products in (3, 1):
        def ws(k):
            H.conv_ws(x, wimgs[k], z, cls, in_scale=s, x_amax=ax, products=products)
        t = timed(ws)
        z.zero_(); ws(0)
        err = float((z - ref).abs().max() / ref.abs().max())
        print(f'      conv_ws products {products}: {t:6.1f} us  diff {err:.1e}', flush=True)
