"""Run-to-run spread of a few accumulating library calls: each is run REPS times on the same inputs and the number of distinct result bit
patterns is printed.  Normal build: more than one for most (float atomics); deterministic build (EG3D_DETERMINISTIC=1): exactly one."""
import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), '3dgan-inversion_amd'))
import torch
from inv3d_amd import hipops as H, _lib as L
dev = torch.device('cuda')
torch.manual_seed(0)
REPS = 12

def distinct(fn):
    outs = []
    for _ in range(REPS):
        r = fn()
        r = r if isinstance(r, (tuple, list)) else (r,)
        outs.append(tuple(t.detach().clone() for t in r))
    torch.cuda.synchronize()
    n = 1
    for o in outs[1:]:
        if not all(torch.equal(a, b) for a, b in zip(o, outs[0])): n += 1
    return n

N, C, Hh, Ww = 1, 128, 256, 256
z = torch.randn(N, C, Hh, Ww, device=dev).contiguous(memory_format=torch.channels_last)
x = torch.randn(N, C, Hh, Ww, device=dev).contiguous(memory_format=torch.channels_last)
s = torch.randn(N, C, device=dev)
def f_dgrad_finish():
    dx = torch.empty_like(z); ds = torch.zeros(N, C, device=dev)
    H.dgrad_finish(z, x, s, dx, ds=ds)
    return ds
print('dgrad_finish ds: runs differing from the first:', distinct(f_dgrad_finish) - 1)

out = torch.randn(N, C, Hh, Ww, device=dev).contiguous(memory_format=torch.channels_last)
nz = torch.randn(Hh, Ww, device=dev); ns = torch.tensor(0.3, device=dev); d = torch.rand(N, C, device=dev) + 0.5; b = torch.randn(C, device=dev)
def f_epi_bwd():
    dz = torch.empty_like(z); dbias = torch.zeros(C, device=dev); dd = torch.zeros(N, C, device=dev); dnoise = torch.zeros(Hh, Ww, device=dev)
    dstrength = torch.zeros((), device=dev)
    H.epilogue_bwd(z, out, dz, d=d, noise=nz, noise_nstride=0, noise_strength=ns, bias=b, act='lrelu', alpha=0.2, gain=1.414, clamp=256.0,
                   dbias=dbias, dd=dd, dnoise=dnoise, dnoise_nstride=0, dstrength=dstrength)
    return dbias, dd, dnoise, dstrength
print('epilogue_bwd reductions: runs differing:', distinct(f_epi_bwd) - 1)

a = torch.randn(200000, 64, device=dev); bb = torch.randn(200000, 32, device=dev)
print('rows_gram: runs differing:', distinct(lambda: H.rows_gram(a, bb)) - 1)

from inv3d_amd import loss_nets as LN
fa = torch.randn(1, 1 << 20, device=dev); fb = torch.randn(1, 1 << 20, device=dev)
print('sqdist: runs differing:', distinct(lambda: LN.sqdist(fa, fb)) - 1)
print('deterministic build:', bool(L.lib().eg3d_det_enabled()), 'misses:', L.det_misses())
if L.lib().eg3d_det_enabled():
    # a call whose targets exceed the lent workspace fails loudly (EG3D_ERR_WORKSPACE), it does not fall back to float atomics
    small = torch.zeros(4096 // 8 + 64, dtype=torch.int64, device=dev)
    L.check(L.lib().eg3d_det_set_workspace(small.data_ptr(), 4096 + 64 * 8, L.stream_ptr()), 'det_set_workspace')
    try:
        f_dgrad_finish()
        print('workspace check: NOT refused')
    except L.Eg3dHipError as e:
        print('workspace check: refused' if 'status -4' in str(e) else f'workspace check: wrong error {e}')
    L._det_ws = None
    L.det_enable()
    print('after restoring the workspace: runs differing:', distinct(f_dgrad_finish) - 1, 'misses:', L.det_misses())

# ---- the accumulator itself: n values of wildly different magnitude and sign, one thread each, in shuffled orders -> the exact sum -------------
import math, random
random.seed(7)
vals = []
for k in range(200000):
    e = random.choice([-60, -40, -20, -10, -3, 0, 5, 12, 20, 30])
    vals.append(random.uniform(-1, 1) * 2.0 ** e)
vals += [3.0e9, -3.0e9, 1.0e-30, -7.0e-38, 1.0e-41]                      # cancellation of large terms, subnormals
vt = torch.tensor(vals, dtype=torch.float32)
exact = math.fsum(float(x) for x in vt.tolist())                          # exact sum of the fp32 values (Shewchuk), rounded once to double
want = torch.tensor(exact, dtype=torch.float64).float()
outs = []
for rep in range(6):
    perm = torch.randperm(vt.numel())
    dv = vt[perm].to(dev)
    tgt = torch.zeros(1, device=dev)
    L.check(L.lib().eg3d_det_accumulate(dv.data_ptr(), dv.numel(), tgt.data_ptr(), L.stream_ptr()), 'det_accumulate')
    outs.append(tgt.cpu())
ulp = abs(float(torch.nextafter(want, torch.tensor(float('inf'))) - want))
err_ulps = max(abs(float(o) - float(want)) / ulp for o in outs)
print(f'accumulator: {len(set(float(o) for o in outs))} distinct result(s) over 6 shuffles, error vs the exact sum {err_ulps:.2f} ulp (exact {exact:.9e})')
