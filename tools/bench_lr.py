"""Stand-alone timing of the low-resolution conv (csrc/conv_lr.hip) against the split-K implicit GEMM + finishing pass it replaces, on the
backbone's 512 -> 512 3x3 layers at 4^2 .. 64^2, with COLD weights (each launch of a replayed graph reads another copy of the weights:
inside the step every layer's 9.4 MB arrive from HBM).   python tools/bench_lr.py [res ...]"""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import hipops as H, _lib as L, fused as F
dev = torch.device('cuda')
NW = 12
ress = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32, 64]
ci = co = 512
g = torch.Generator().manual_seed(0)
ws = [(torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)).to(dev) for _ in range(NW)]
wfs = [H.pack_weight_fwd(w) for w in ws]
wimgs = [H.split_weight(wf, co, ci, 9) for wf in wfs]
wps = [H.split_weight_pieces(wf) for wf in wfs]
s = (1 + 0.5 * torch.randn(1, ci, generator=g)).to(dev)
d = (0.5 + torch.rand(1, co, generator=g)).to(dev)
bias = torch.zeros(co, device=dev)
strength = torch.tensor(0.1, device=dev)


def timed(fn, reps=5):
    """fn(k) issues launch k of NW; captured once, replayed `reps` times; returns us per launch."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for k in range(NW):
            fn(k)
        st.synchronize()
        gr = torch.cuda.CUDAGraph()
        with H.capture_guard(), torch.cuda.graph(gr, stream=st):
            for k in range(NW):
                fn(k)
        gr.replay(); st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record(st); gr.replay(); e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / NW)
    return best


for res in ress:
    x = torch.randn(1, ci, res, res, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ax = H.absmax(x)
    noise = torch.randn(res, res, generator=g).to(dev)
    cls = H.classes_corr(res, res, 3, 3, 1)
    out = H.empty_cl(1, co, res, res, dev)
    amax = torch.zeros(1, device=dev)
    epi = dict(noise=noise, noise_nstride=0, noise_strength=strength, bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=-1.0)
    ks_old = F._auto_ksplit(cls, 1, co, ci)
    z = torch.zeros(1, co, res, res, device=dev).contiguous(memory_format=torch.channels_last)

    def old(k):
        z.zero_()
        if ks_old > 1:
            H.conv_atomic(x, wfs[k], ci, co, z, cls, in_scale=s, ksplit=ks_old, precision='f16x3', w_pieces=wps[k])
            H.epilogue_fwd(z, out, d=d, out_amax=amax, **epi)
        else:
            H.conv_igemm(x, wfs[k], ci, co, out, cls, in_scale=s, epi=L.EPI_FWD, out_scale=d, precision='f16x3', out_amax=amax, w_pieces=wps[k], **epi)
    t_old = timed(old)
    print(f'res {res:3d}: igemm split-K {ks_old:2d} + fill + finishing pass {t_old:7.1f} us', flush=True)
    plan0 = H.conv_lr_plan(ci, co, cls, 1, force=True)
    for rpw in (1, 2, 4):
        if H.conv_lr_plan(ci, co, cls, 1, force=True, rpw=rpw)[2] != rpw or rpw * 64 > 2 * res * res:
            continue
        for ks in (1, 2, 4, 8, 16):
            def new(k):
                H.conv_lr(x, ax, wimgs[k], out, cls, (plan0[0], ks, rpw), in_scale=s, epi=L.EPI_FWD, out_scale=d, out_amax=amax, **epi)
            try:
                t = timed(new)
            except Exception as e:       # noqa: BLE001
                print('   lr ks', ks, 'failed:', str(e)[:80]); continue
            print(f'          conv_lr logw {plan0[0]} cells {64 * rpw:3d} ks {ks:2d}: {t:7.1f} us' + ('   <- plan' if (ks, rpw) == plan0[1:] else ''), flush=True)
