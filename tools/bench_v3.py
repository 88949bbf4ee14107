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

