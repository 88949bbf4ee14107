#!/usr/bin/env python3
"""bench.py -- inversion-steps/sec of the MI355X-native EG3D inversion inner loop (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one Phase-A latent-inversion step of config C2 (BASELINE.json configs[1], "FFHQ 512^2
single-image w+ inversion, 1 x MI355X"): G.synthesis forward with grad on the full-size ffhqrebalanced512-128-shaped generator
(30.66 M params, 128^2 x (48+48)-sample neural rendering, 512^2 output), feature-space distance + noise regulariser, backward
into the latent and the 13+4 noise buffers, Adam step, noise renormalisation.  Weights/latents/cameras are synthetic
(deterministic per-tensor generator); the perceptual net is the stub feature pyramid of inv3d_amd.inversion.
N GPUs = N independent images (weak scaling), one packed stat all-reduce per step over RCCL.

The JSON line also carries
  roofline      : the dominant kernel = the conv kernel family (tile configuration x arithmetic) with the largest share of the step's
                  time: conv_v2_kernel<9,true> (csrc/conv_v2.hip, pre-split fp16 pieces, LDS-DMA staged halo) on the default settings.
                  achieved = algorithmic FLOPs of its launches (2 x MACs of the convolution) / their HIP-event
                  durations measured on the launch stream.  `peak` is the guide's peak of the MFMA the kernel issues
                  (MI355X_MICROARCH.md: 2500 TFLOP/s dense for v_mfma_f32_32x32x16_f16 / bf16, 157.3 for v_mfma_f32_32x32x2_f32) and
                  `frac` = achieved / peak.  The split modes form every fp32 product from `products_per_fp32_product` exact 16-bit products
                  (3 fp16 in the default mode, 6 bf16 with --precision bf16x6): `mfma_executed_tflops` / `frac_executed` count those.
                  `families` lists every conv kernel family of the step with its own fraction, `furthest_from_roofline` names the one
                  with the lowest (>= 5 % of the conv time), `step_tflops` / `step_frac` = 611.6 GFLOP / ms_per_step against 2500,
                  `roofline_backbone` = SURVEY 8d's backbone-only figure (93.1 GF x 2 passes / backbone time / 2500), the measured
                  register-only MFMA rate and the same figures over ALL conv launches (`all_conv_*`) are beside it; `traffic` = HBM bytes
                  per launch from the committed PMC passes (profiles/traffic_table.json).
  roofline_renderer : the volume renderer's forward + backward against HBM (SURVEY section 8d algorithmic bytes).
  final_psnr    : one image through the whole 400 + 400 step budget (InversionCoach): the second half of the metric.
  cpu_baseline  : the CPU oracle (oracle/eg3d_oracle.py, a port of the reference's pure-PyTorch `_ref` path, pinned against
                  the reference) running the same C2 step on the host cores, bounded sample, rank 0 at N=1 only.
  cpu_baseline_c1 : the oracle's pivotal-tuning step at full size on the same cores (BASELINE.json configs[0]: 10 PTI steps on CPU).
  side_configs  : w+ latent, VGG16-LPIPS architecture, 8 images per GPU (config C5's per-GPU share), the pivotal-tuning step (config C4)
                  and the plain `G.synthesis` loop of a drop-in caller, each timed for a few steps in the same run (never the headline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, '3dgan-inversion_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide, dense bf16 (never the 2:1-sparsity figure)
PRODUCTS = {'f32': 1, 'bf16x6': 6, 'bf16x3': 3, 'f16x3': 3}      # MFMA products executed per algorithmic fp32 product
TRAFFIC_TABLE = os.path.join(ROOT, 'profiles', 'traffic_table.json')     # per-kernel HBM bytes per launch from the committed rocprofv3 PMC passes


def cpu_baseline_c2(seconds_budget=30.0):
    """Time the oracle's C2 step (same generator shape, same loss structure: oracle/inversion_oracle.ProjectorOracle) on the host."""
    from oracle import eg3d_oracle as O
    from oracle import inversion_oracle as IO
    threads = min(os.cpu_count() or 1, 32)        # more threads than this only adds contention for these op sizes
    torch.set_num_threads(threads)
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    c = O.synth_cameras(1, seed=2)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    proj = IO.ProjectorOracle(P, cfg, target, num_steps=400, cam=c, w_start=O.synth_ws(cfg, 1, seed=1)[:, :1])
    t0 = time.time()
    proj.step(u1, u2)                       # warm-up
    warm = time.time() - t0
    times = []
    while len(times) < 3 and (time.time() - t0) < seconds_budget:
        t1 = time.time()
        proj.step(u1, u2)
        times.append(time.time() - t1)
    if not times:
        times = [warm]
    times.sort()
    med = times[len(times) // 2]
    return dict(value=round(1.0 / med, 4), unit='steps/s', cores=threads, kind='port',
                sample=f'oracle C2 step (ProjectorOracle), full-size generator, N=1: 1 warm-up + {len(times)} timed steps, median {med:.2f} s/step')


def self_launch_command(gpus, argv, port=None):
    """The command `bench.py --gpus N` re-launches itself with when it was not started by a launcher: the driver's own form
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(gpus)}', '--master-addr', '127.0.0.1',
            '--master-port', str(int(port)), os.path.abspath(__file__)] + list(argv)


def cpu_baseline_c1(seconds_budget=40.0):
    """BASELINE.json configs[0] ("1 image, 10 PTI steps on CPU via the pure-Python fallback"): the pivotal-tuning step of the oracle
    (PivotalTunerOracle: forward, objective, backward into all 30.7 M weights, Adam) on the full-size generator, bounded sample."""
    from oracle import eg3d_oracle as O
    from oracle import inversion_oracle as IO
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    cfg = O.full_config()
    P = O.synth_params(cfg, seed=0)
    c = O.synth_cameras(1, seed=2)
    u1, u2 = O.make_uniforms(cfg, 1, seed=4)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    tuner = IO.PivotalTunerOracle(P, cfg, target, O.synth_ws(cfg, 1, seed=1), c)
    t0 = time.time()
    tuner.step(u1, u2, noise_mode='const')          # warm-up
    times = []
    while len(times) < 2 and (time.time() - t0) < seconds_budget:
        t1 = time.time()
        tuner.step(u1, u2, noise_mode='const')
        times.append(time.time() - t1)
    if not times:
        times = [time.time() - t0]
    med = sorted(times)[len(times) // 2]
    return dict(value=round(1.0 / med, 4), unit='steps/s', cores=threads, kind='port', ten_pti_steps_s=round(10 * med, 1),
                sample=f'oracle pivotal-tuning step (PivotalTunerOracle), full-size generator, N=1: 1 warm-up + {len(times)} timed steps, median {med:.2f} s/step; '
                       'config C1 (10 PTI steps) = 10 x that')


def measure_mfma_probe(dev):
    """TFLOP/s a register-only v_mfma_f32_32x32x16_f16 loop sustains on random data, measured now (eg3d_probe_mfma_f16, HIP events)."""
    from inv3d_amd import _lib as L
    data = (torch.rand(4096 * 8, device=dev) * 2 - 1).mul_(1000.0).half()
    blocks, iters = 1024, 2000
    out = torch.empty(blocks * 256, device=dev)
    call = lambda: L.check(L.lib().eg3d_probe_mfma_f16(data.data_ptr(), out.data_ptr(), blocks, iters, L.stream_ptr()), 'probe_mfma_f16')   # noqa: E731
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return blocks * 4 * iters * 24 * 32768.0 / (ms * 1e-3) / 1e12


def measure_backbone(G, dev, M, reps=20):
    """GPU time of the StyleGAN2 backbone alone, forward and backward, each replayed from its own HIP graph (an eager span would be
    host-bound): backbone.synthesis(ws) -> planes, then d planes -> (d ws, d noise maps) with the weights frozen, as in the C2 step."""
    from inv3d_amd import hipops as H
    from inv3d_amd import synthetic as S
    ws = S.synth_ws(14, 512, M, seed=7).to(dev).requires_grad_(True)
    bufs = [b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n]
    was = [b.requires_grad for b in bufs]
    for b in bufs:
        b.requires_grad = True
    arena_f, arena_b = H.ZeroArena(dev), H.ZeroArena(dev)
    side = torch.cuda.Stream(device=dev)
    try:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                         # warm-up: weight images, arena sizes
                with H.zero_arena(arena_f):
                    planes = G.backbone.synthesis(ws, noise_mode='const')
                g = torch.randn_like(planes)
                with H.zero_arena(arena_b):
                    torch.autograd.grad(planes, [ws] + bufs, g, allow_unused=True)
            side.synchronize()
            fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with H.capture_guard(), torch.cuda.graph(fwd, stream=side, capture_error_mode='thread_local'), H.zero_arena(arena_f):
                planes = G.backbone.synthesis(ws, noise_mode='const')
            with H.capture_guard(), torch.cuda.graph(bwd, pool=fwd.pool(), stream=side, capture_error_mode='thread_local'), H.zero_arena(arena_b):
                grads = torch.autograd.grad(planes, [ws] + bufs, g, allow_unused=True)
            fwd.replay(); bwd.replay(); side.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            for _ in range(reps):
                ev[0].record(side); fwd.replay(); ev[1].record(side); bwd.replay(); ev[2].record(side)
                side.synchronize()
                tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        del grads, planes, fwd, bwd
        return tf / reps, tb / reps
    finally:
        for b, w in zip(bufs, was):
            b.requires_grad = w
        torch.cuda.current_stream().wait_stream(side)


def _time_steps(fn, n, warm, repeats=1, spread=None):
    """ms per call: `repeats` timed regions of n calls each, the MEDIAN region; spread (a dict) receives min / median / max / repeats."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(max(1, repeats)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    ts.sort()
    if spread is not None:
        spread.update(repeats=len(ts), calls_per_region=n, ms_min=round(ts[0], 3), ms_median=round(ts[len(ts) // 2], 3), ms_max=round(ts[-1], 3))
    return ts[len(ts) // 2]


def side_configs(G, target, cam, dev, use_graph, steps=10):
    """The other configurations of BASELINE.json / SURVEY section 8d, measured in the same run as side figures (never the headline):
    w+ latent, the VGG16-LPIPS architecture in the loop, 8 images per GPU (the per-GPU share of config C5), the pivotal-tuning step
    (config C4) and the PLAIN `G.synthesis` loop a drop-in caller runs (no projector object: inv3d_amd/graphed.py)."""
    from inv3d_amd import synthetic as S
    from inv3d_amd.inversion import LatentProjector, PivotalTuner
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:           # a side figure must never cost the benchmark line
            out[name] = dict(error='%s: %s' % (type(e).__name__, e))
        torch.cuda.synchronize()

    def projector(**kw):
        def run():
            pr = LatentProjector(G, kw.pop('target', target), num_steps=400, cam=kw.pop('cam', cam), seed=100, use_graph=use_graph, **kw)
            pr.preheat = 0
            m = pr.N
            sp = {}
            ms = _time_steps(pr.step, steps, pr._graph_warmup + 2, repeats=5, spread=sp)
            if use_graph and pr._graph is None:
                raise RuntimeError(f'capture failed: {pr.graph_capture_error}')
            return dict(ms_per_step=round(ms, 3), image_steps_per_s=round(m * 1e3 / ms, 2), spread=sp)
        return run

    guarded('wplus', projector(wplus=True))

    def vgg():
        from inv3d_amd.loss_nets import VGG16LPIPS
        return projector(feature_net=VGG16LPIPS().to(dev))()
    guarded('loss_net_vgg16', vgg)

    def c5():
        m = 8
        cams = S.synth_cameras(m, seed=2).to(dev)
        with torch.no_grad():
            ws_t = S.synth_ws(14, 512, m, seed=3).to(dev)
            tg = torch.cat([G.synthesis(ws_t[i:i + 1], cams[i:i + 1], noise_mode='const', force_fp32=True)['image'].clamp(-1, 1) for i in range(m)])
        r = projector(target=tg, cam=cams)()
        r['note'] = '8 independent inversions as one batch on this GPU = the per-GPU share of config C5 (64 images on 8 GPUs)'
        # the conv kernels with the chip filled (at one image the 4^2 .. 128^2 layers are latency-bound): same accounting as `roofline`
        from inv3d_amd import hipops as H
        eager = LatentProjector(G, tg, num_steps=400, cam=cams, seed=100, use_graph=False)
        eager.preheat = 0
        eager.step()
        prof = H.LaunchProfiler()
        H.PROFILER = prof
        try:
            for _ in range(2):
                eager.step()
            torch.cuda.synchronize()
        finally:
            H.PROFILER = None
        summ = prof.summary()
        ms, fl = sum(v['ms'] for v in summ.values()), sum(v['flops'] for v in summ.values())
        if ms > 0:
            r['all_conv_tflops'] = round(fl / (ms * 1e-3) / 1e12, 1)
            r['all_conv_frac'] = round(fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)            # algorithmic TFLOP/s / 2500 (the MFMAs issued are 16-bit)
            r['all_conv_frac_executed'] = round(3 * fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)
            r['all_conv_ms_per_step'] = round(ms / 2, 3)
        try:        # SURVEY 8d's backbone-only figure with the chip filled: the north star's "MFMA on the backbone" is a batch question (at one image 0.039)
            bf_ms, bb_ms = measure_backbone(G, dev, m)
            tf = 93.1 * 2 * m * 1e9 / ((bf_ms + bb_ms) * 1e-3) / 1e12
            r['roofline_backbone'] = dict(gflop_algorithmic=round(93.1 * 2 * m, 1), fwd_ms=round(bf_ms, 3), bwd_ms=round(bb_ms, 3), tflops=round(tf, 1),
                                          peak=BF16_MFMA_PEAK_TFLOPS, frac=round(tf / BF16_MFMA_PEAK_TFLOPS, 4), frac_executed=round(3 * tf / BF16_MFMA_PEAK_TFLOPS, 4))
        except Exception as e:
            r['roofline_backbone'] = dict(error='%s: %s' % (type(e).__name__, e))
        return r
    guarded('images_per_gpu_8', c5)

    def plain_loop():
        # the shape of training/projectors/w_projector.py:189-261 with no object of this package around the call
        import torch.nn.functional as F
        G.requires_grad_(False)
        bufs = [b for n, b in G.backbone.synthesis.named_buffers() if 'noise_const' in n]
        for b in bufs:
            b.requires_grad = True
        saved = [b.detach().clone() for b in bufs]
        w_opt = S.synth_ws(14, 512, 1, seed=1)[:, :1].to(dev).clone().requires_grad_(True)
        opt = torch.optim.Adam([w_opt] + bufs, lr=0.01, fused=True)
        t256 = F.avg_pool2d(target[:1], 2)

        def step():
            ws = (w_opt + 0.01 * torch.randn_like(w_opt)).repeat(1, 14, 1)
            o = G.synthesis(ws, cam[:1], noise_mode='const', force_fp32=True)
            loss = (F.avg_pool2d(o['image'], 2) - t256).square().sum()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        try:
            res = {}
            for flag, key in ((True, 'ms_per_step'), (False, 'ms_per_step_per_launch_path')):
                G.graph_eager = flag
                res[key] = round(_time_steps(step, steps, 4), 3)
            res['note'] = 'plain `out = G.synthesis(ws, c, noise_mode="const", force_fp32=True); loss.backward(); opt.step()` loop, L2 loss at 256^2: forward and backward replayed from HIP graphs inside G.synthesis vs one launch per kernel'
            return res
        finally:
            G.graph_eager = True
            with torch.no_grad():
                for b, v in zip(bufs, saved):
                    b.requires_grad = False
                    b.copy_(v)
    guarded('plain_g_synthesis_loop', plain_loop)

    def phase_a_sr_f16x1():
        # SURVEY section 7 / VERDICT r2 item 10: Phase A with the SR head in the reference's fp16-operand arithmetic (one MFMA product instead of
        # three) -- the whole 400-step latent projection twice from the same seed, final PSNR of both and the drift between them
        from inv3d_amd.inversion import psnr_01
        res = {}
        for key, flag in (('f16x3', False), ('sr_f16x1', True)):
            pr = LatentProjector(G, target[:1], num_steps=400, cam=cam[:1], seed=321, use_graph=use_graph, sr_fp16=flag)
            pr.preheat = 0
            for _ in range(pr._graph_warmup + 1):
                pr.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_timed = 400 - pr.step_idx
            for _ in range(n_timed):
                out = pr.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[key] = dict(steps_per_s=round(n_timed / dt, 2), final_psnr_db=round(float(psnr_01(out['image'], target[:1])), 4))
        res['final_psnr_drift_db'] = round(abs(res['f16x3']['final_psnr_db'] - res['sr_f16x1']['final_psnr_db']), 4)
        res['note'] = '400-step latent projection, same seed; sr_f16x1 = super-resolution head with one product of fp16-rounded operands (the reference uses force_fp32=True in this phase: an option, not the default)'
        return res
    guarded('phase_a_sr_f16x1', phase_a_sr_f16x1)

    def ref_arith_f16x1():
        # the arithmetic class of the reference's own GPU path, labelled.  The reference's inversion scripts never disable
        # TF32 (only training/training_loop.py:135-136 and calc_metrics.py:52-53 do), so on the RTX 3090 it names every cuDNN conv of
        # G.synthesis keeps an 11-bit significand per operand -- what ONE product of range-normalised fp16-rounded operands keeps.  Here: every
        # modulated conv that runs on the pre-split kernels (backbone AND super-resolution head) in one product, fp32 accumulation and results.
        from inv3d_amd.inversion import psnr_01
        from inv3d_amd import hipops as H
        import torch.nn.functional as F
        res = {}
        ws1 = S.synth_ws(14, 512, 1, seed=7).to(dev)
        imgs, grads = {}, {}
        ge = G.graph_eager
        G.graph_eager = False              # two one-off passes: launched kernel by kernel, nothing captured for them
        try:
            for key, ov in (('f16x3', None), ('f16x1', 'f16x1')):
                w = ws1.clone().requires_grad_(True)
                with H.modconv_override(ov):
                    o = G.synthesis(w, cam[:1], noise_mode='const', force_fp32=True)
                    gw, = torch.autograd.grad((F.avg_pool2d(o['image'], 2) - F.avg_pool2d(target[:1], 2)).square().sum(), [w])
                imgs[key], grads[key] = o['image'].detach().clone(), gw.detach().clone()
                del o, gw, w
        finally:
            G.graph_eager = ge
        torch.cuda.synchronize()
        res['forward_psnr_vs_f16x3_db'] = round(float(psnr_01(imgs['f16x1'], imgs['f16x3'])), 2)
        res['forward_max_abs_err'] = float((imgs['f16x1'] - imgs['f16x3']).abs().max())
        res['d_ws_rel_err'] = float((grads['f16x1'] - grads['f16x3']).abs().max() / grads['f16x3'].abs().max())
        for key, flag in (('f16x3', False), ('f16x1', True)):
            pr = LatentProjector(G, target[:1], num_steps=400, cam=cam[:1], seed=321, use_graph=use_graph, modconv_f16x1=flag)
            pr.preheat = 0
            for _ in range(pr._graph_warmup + 1):
                pr.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_timed = 400 - pr.step_idx
            for _ in range(n_timed):
                out = pr.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[key] = dict(steps_per_s=round(n_timed / dt, 2), final_psnr_db=round(float(psnr_01(out['image'], target[:1])), 4))
        res['final_psnr_drift_db'] = round(abs(res['f16x3']['final_psnr_db'] - res['f16x1']['final_psnr_db']), 4)
        res['headline_arithmetic'] = 'f16x3 (this entry is a side figure: the headline moves to one product only if the 400-step drift is <= 1e-3 dB)'
        res['note'] = ('every modulated conv on the pre-split kernels (backbone + SR head; conv_v2 / conv_v3 / up2 / s2adj) in ONE v_mfma_f32_32x32x16_f16 product of '
                       'range-normalised fp16-rounded operands, fp32 accumulation / results; layers still on the loader-split kernel keep three products.  The reference '
                       'runs TF32 convs on its inversion path on its named GPU (TF32 is only disabled in training/training_loop.py:135-136 and calc_metrics.py:52-53): '
                       'same 11-bit operand significand.  forward_psnr / d_ws_rel_err: one full-size forward + C2-style backward against the three-product path; '
                       'final_psnr_drift: two 400-step latent projections from the same seed')
        return res
    guarded('ref_arith_f16x1', ref_arith_f16x1)

    def phase_b():
        import copy
        state = copy.deepcopy(G.state_dict())
        try:
            w_pivot = S.synth_ws(14, 512, 1, seed=5).to(dev)
            tuner = PivotalTuner(G, target[:1], w_pivot, cam[:1], use_graph=use_graph)
            sp = {}
            ms = _time_steps(tuner.step, steps, 4, repeats=5, spread=sp)
            if use_graph and tuner._graph is None:
                raise RuntimeError(f'capture failed: {tuner.graph_capture_error}')
            return dict(ms_per_step=round(ms, 3), steps_per_s=round(1e3 / ms, 2), spread=sp,
                        note='config C4: pivotal-tuning step, all 30.7 M weights trainable (forward + data and weight gradients + fused Adam), SR head in the reference\'s fp16-operand arithmetic')
        finally:
            G.load_state_dict(state)
            G.requires_grad_(False)
            from inv3d_amd import hipops as H
            H.weights_changed()
    guarded('phase_b_c4', phase_b)

    def c3(real_nets):
        # BASELINE.json configs[2]: C2 + pose optimisation (free quaternion | the ResNet-34 pose estimator fine-tuned in the loop) + translation
        # + a second, no-grad forward at the canonical camera + the depth-reprojection warping loss (w_projector.py:145-270, warping_loss.py:6-72)
        def run():
            kw = {}
            if real_nets:
                from inv3d_amd.loss_nets import VGG16LPIPS, VGG16Features
                from inv3d_amd.pose_net import resnet34_pose
                kw = dict(pose_net=resnet34_pose(4).to(dev), feature_net=VGG16LPIPS().to(dev), warp_feature_net=VGG16Features().to(dev))
            pr = LatentProjector(G, target, num_steps=400, optimize_pose=True, use_warping_loss=True, cam_preheat_steps=2, seed=1, use_graph=use_graph, **kw)
            ms = _time_steps(pr.step, steps, pr._graph_warmup + 4)          # the two camera-preheat steps run eagerly, the steady-state step is captured
            if use_graph and pr._graph is None:
                raise RuntimeError(f'capture failed: {pr.graph_capture_error}')
            return dict(ms_per_step=round(ms, 3), steps_per_s=round(1e3 / ms, 2),
                        gflop_algorithmic=917.4, tflops=round(917.4e9 / (ms * 1e-3) / 1e12, 1),
                        note=('config C3: latent + pose (%s) + translation, canonical-view forward, warping loss; %s; three optimisers in the step'
                              % (('ResNet-34 pose estimator fine-tuned in the loop', 'VGG16-LPIPS + VGG16 features[:15] architectures (random weights)') if real_nets
                                 else ('free quaternion', 'stub feature nets'))))
        return run
    guarded('c3_pose_warp', c3(False))
    guarded('c3_pose_warp_real_nets', c3(True))

    def deterministic():
        # The deterministic build (csrc/det.h: every floating-point atomic an exact fixed-point accumulation) is chosen when the library is
        # loaded, so its cost is timed in a fresh interpreter: the same C2 loop (graph replay), two runs compared bit for bit.
        import subprocess
        env = dict(os.environ, EG3D_DETERMINISTIC='1')
        env.pop('EG3D_LIBNAME', None)
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'det_runs.py'), 'c2', '60', 'graph'],
                           env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-400:])
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return dict(workload='C2 under EG3D_DETERMINISTIC=1 (libeg3d_hip_det.so), HIP-graph replay', ms_per_step=d['ms_per_step'],
                    steps_per_s=round(1e3 / d['ms_per_step'], 2), two_runs_bit_identical=bool(d['equal']), float_atomic_fallbacks=d['misses'], steps=d['steps'])
    guarded('deterministic_c2', deterministic)
    return out


def final_psnr_run(G, target, cam, use_graph, feature_net):
    """Second half of the metric: one image through the whole budget of configs/hyperparameters.py (400 latent + 400 pivotal-tuning steps)."""
    from inv3d_amd.coach import InversionCoach
    t1 = time.perf_counter()
    try:
        coach = InversionCoach(G, first_inv_steps=400, max_pti_steps=400, lpips_threshold=0.0, use_graph=use_graph, early_stop_interval=1, w_avg_samples=0,
                               feature_net=feature_net)
        res = coach.invert('bench', target[:1], cam[:1])
        torch.cuda.synchronize()
        modes = coach.last_launch_modes                 # how the two phases were actually issued: stated in the note (a refused capture falls back to eager launches)
        return dict(final_psnr_db=round(res.psnr_tuned, 3), pivot_psnr_db=round(res.psnr_pivot, 3), steps=res.steps_a + res.steps_b,
                    wall_s=round(time.perf_counter() - t1, 2), note='400 latent steps (fp32-equivalent) + 400 pivotal-tuning steps (SR head in the reference\'s fp16-operand arithmetic, as BaseCoach.forward), %s, stub feature pyramid, synthetic target' % ('both phases replayed from HIP graphs (the early-stop criterion is evaluated on the device in every step of the captured tuning step; the host polls the flag every step)' if modes == dict(phase_a='graph', phase_b='graph') else 'launch modes: %s' % modes))
    except Exception as e:           # the side run must never cost the benchmark line
        return dict(error='%s: %s' % (type(e).__name__, e))


def run_child(kind, args, timeout=600):
    """`bench.py --child side|final` with the parent's workload flags; returns the child's dictionary or an error entry."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--child', kind, '--no-cpu-baseline', '--no-roofline', '--loss-net', args.loss_net]
    if args.no_graph:
        cmd.append('--no-graph')
    if args.precision is not None:
        cmd += ['--precision', args.precision]
    if args.wplus:
        cmd.append('--wplus')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return dict(error=f'child process ({kind}) exceeded {timeout} s')
    for ln in reversed(r.stdout.strip().splitlines()):
        if ln.startswith('{"child"'):
            return json.loads(ln)['child']
    return dict(error=f'child process ({kind}) ended with code {r.returncode} and no result', stderr_tail=r.stderr.strip()[-300:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each; the line reports the median region and the spread')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel from the host instead of replaying the captured step')
    ap.add_argument('--precision', default=None, choices=sorted(PRODUCTS) + ['auto'],
                    help="matrix-core arithmetic of the implicit GEMMs (default 'auto': f16x3 for the modulated convs, bf16x6 elsewhere)")
    ap.add_argument('--wplus', action='store_true')
    ap.add_argument('--images-per-gpu', type=int, default=1,
                    help='images inverted as one batch on every GPU (1 = config C2; 8 = the per-GPU share of config C5: 64 images on 8 GPUs)')
    ap.add_argument('--no-final-psnr', action='store_true', help='skip the full-budget (400 + 400 steps) inversion that reports the final PSNR')
    ap.add_argument('--no-side-configs', action='store_true', help='skip the side figures (w+, VGG16-LPIPS, 8 images per GPU, pivotal tuning, plain G.synthesis loop)')
    ap.add_argument('--loss-net', default='stub', choices=['stub', 'vgg16'],
                    help="feature network of the LPIPS term: 'stub' = the small fixed conv pyramid the C2 workload is defined with (SURVEY.md "
                         "section 8d); 'vgg16' = the full VGG16-LPIPS architecture (random weights) on the same kernels")
    ap.add_argument('--child', default=None, choices=['side', 'final'], help=argparse.SUPPRESS)      # internal: the side figures / the final-PSNR run in a process of their own
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # not under torch.distributed.run: launch the ranks ourselves (one process per GPU, RCCL rendezvous on the loopback address)
        import subprocess
        sys.exit(subprocess.call(self_launch_command(args.gpus, sys.argv[1:])))

    from inv3d_amd import dist as D
    # EG3D_BENCH_BACKEND=gloo + EG3D_BENCH_ONE_DEVICE=1: dry-run of the multi-rank control flow (async stat reducer, barriers, rank-max timing)
    # with every rank on GPU 0 of a single-GPU box -- RCCL refuses two ranks on one device; never used for a reported number
    if os.environ.get('EG3D_BENCH_ONE_DEVICE') == '1':
        os.environ['LOCAL_RANK'] = '0'
    rank, world, local = D.init_from_env(os.environ.get('EG3D_BENCH_BACKEND', 'nccl'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback in the product path)'
    assert world == args.gpus, f'--gpus {args.gpus} but the launcher started {world} rank(s)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cpus = D.pin_to_numa(int(os.environ.get('LOCAL_RANK', local)) if os.environ.get('EG3D_BENCH_ONE_DEVICE') != '1' else rank, world)   # each rank on its GPU's NUMA cores
    D.warm_up(dev)              # RCCL builds its communicator inside the first collective: before any capture (dist.assert_comm_ready)

    from inv3d_amd import synthetic as S, hipops as H
    if args.precision is not None:
        H.set_conv_precision(args.precision)
    prec_name = H.modconv_precision()          # arithmetic of the dominant kernel's launches
    from inv3d_amd.inversion import LatentProjector, psnr_01

    G = S.make_generator(device=dev)
    S.load_synthetic_weights(G, seed=0)
    # per-rank independent images: targets = renders of different latents by the same generator (so PSNR is meaningful)
    M = args.images_per_gpu
    cam = S.synth_cameras(world * M, seed=2)[rank * M:(rank + 1) * M].to(dev)
    with torch.no_grad():
        ws_t = S.synth_ws(14, 512, world * M, seed=3)[rank * M:(rank + 1) * M].to(dev)
        target = torch.cat([G.synthesis(ws_t[i:i + 1], cam[i:i + 1], noise_mode='const', force_fp32=True)['image'].clamp(-1, 1) for i in range(M)])
    use_graph = not args.no_graph
    feature_net = None
    if args.loss_net == 'vgg16':
        from inv3d_amd.loss_nets import VGG16LPIPS
        feature_net = VGG16LPIPS().to(dev)
    if args.child == 'side':             # (same generator, target and camera as the parent built: everything here is seeded)
        print(json.dumps(dict(child=side_configs(G, target, cam, dev, use_graph))), flush=True)
        return
    if args.child == 'final':
        print(json.dumps(dict(child=final_psnr_run(G, target, cam, use_graph, feature_net))), flush=True)
        return
    proj = LatentProjector(G, target, num_steps=400, cam=cam, wplus=args.wplus, seed=100 + rank, use_graph=use_graph, feature_net=feature_net)
    proj.preheat = 0

    ones = torch.ones(1, device=dev)
    reducer = D.StepStatReducer(4, dev)      # the path's only collective: packed per-step stats, reduced asynchronously (no rank lockstep)

    def one_step(pr=proj):
        out = pr.step()
        if world > 1:       # summed on the device, read after the timed region
            reducer.push(torch.stack([out['loss'].reshape(()), out['dist'].reshape(()), ones[0], ones[0]]))
        return out

    if use_graph:                       # set-up, not a step of the benchmark: eager passes + the capture of the step into a HIP graph
        D.assert_comm_ready()
        for _ in range(proj._graph_warmup + 1):
            one_step()
        if proj._graph is None:         # a number measured on the eager fallback must not pass for the captured step
            raise RuntimeError(f'HIP graph capture of the step failed: {proj.graph_capture_error}')
    for _ in range(args.warmup):
        one_step()
    def profiler():
        return H.LaunchProfiler()           # every implicit-GEMM launch (grouped by kernel afterwards) + the renderer's forward / backward spans

    prof = None
    if not args.no_roofline and not use_graph:
        prof = profiler()
        H.PROFILER = prof
    # `--repeats` timed regions of EXACTLY `--steps` steps, each bracketed by a device synchronise + barrier on both sides and reduced to the maximum
    # over ranks; the line reports the MEDIAN region (a single 0.08 .. 0.8 s sample on one box cannot tell a +3 % change from a lucky lease) and the
    # min / max beside it.  With the launch profiler on (eager --no-graph runs) one region only: the profiler brackets every launch of it.
    regions = []
    for _ in range(1 if prof is not None else max(1, args.repeats)):
        torch.cuda.synchronize()
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
        D.barrier()
        regions.append(D.max_over_ranks(time.perf_counter() - t0, dev))
    elapsed = sorted(regions)[len(regions) // 2]
    H.PROFILER = None
    step_stats = reducer.finish()              # [sum loss, sum dist, steps x ranks, .] over everything pushed (warm-up included)
    psnr_now = float(psnr_01(proj.last['image'], target))
    roofline_pass = 'HIP events around every launch of the kernel inside the timed region'
    if not args.no_roofline and use_graph:
        # HIP events cannot bracket a kernel inside a captured graph: the per-launch durations of the dominant kernel come from an
        # instrumented EAGER pass over the same K steps (same generator, same shapes), run right after the timed region.
        roofline_pass = 'HIP events around every launch of the kernel in an eager re-run of the same %d steps right after the timed (graph-replay) region' % args.steps
        eager = LatentProjector(G, target, num_steps=400, cam=cam, wplus=args.wplus, seed=100 + rank, use_graph=False, feature_net=feature_net)
        eager.preheat = 0
        one_step(eager)
        prof = profiler()
        H.PROFILER = prof
        for _ in range(args.steps):
            one_step(eager)
        torch.cuda.synchronize()
        H.PROFILER = None
        del eager

    roof = roof_r = None
    probe_tf = None
    if prof is not None:
        try:
            probe_tf = measure_mfma_probe(dev)
        except Exception:
            probe_tf = None
        traffic = json.load(open(TRAFFIC_TABLE)) if os.path.exists(TRAFFIC_TABLE) else {}
        summ = prof.summary()
        dom_id = max(summ, key=lambda k: summ[k]['ms']) if summ else None      # the kernel with the largest share of the step
        dom = summ.get(dom_id)
        if dom and dom['ms'] > 0:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            is_v2 = dom_id[0] in (H.V2_CONFIG, H.V2H_CONFIG, H.V2Q_CONFIG)
            v2_rpw = {H.V2_CONFIG: 4, H.V2H_CONFIG: 2, H.V2Q_CONFIG: 1}.get(dom_id[0], 4)          # patch rows per wave: template argument of the instantiation
            dom_prec = {v: k for k, v in H.PRECISIONS.items()}[dom_id[1]]          # arithmetic of the dominant kernel's launches
            nprod = PRODUCTS[dom_prec]
            peak = FP32_MFMA_PEAK_TFLOPS if dom_prec == 'f32' else BF16_MFMA_PEAK_TFLOPS / nprod
            if is_v2:
                tkey = 'conv_v2_kernel<9,true,false,%d,false,1>' % v2_rpw        # (the name rocprofv3 prints: NTAPS, FULL, ATOMIC, RPW, RGB, KH)
                kern = tkey + ' (pre-split fp16 pieces, LDS-DMA staged halo; fp32 in/out, 3 x v_mfma_f32_32x32x16_f16 per fp32 product)'
            else:
                kern = 'conv_igemm_kernel<%s,%d,*> (fp32 in/out, %d x MFMA per fp32 product)' % (H.TILE_NAMES.get(dom_id[0], '?'), H.PRECISIONS[dom_prec], nprod)
                tkey = 'conv_igemm_kernel<%s>' % H.TILE_NAMES.get(dom_id[0], '?')
            tr = (traffic.get(tkey) or traffic.get(tkey.replace(',1>', '>'))          # (tables written before the KH template argument existed)
                  or traffic.get(tkey.replace(',false,1>', '>')) or traffic.get(tkey.replace(',false>', '>')) or traffic.get(tkey.replace(',true>', '>'), {}))
            per_launch_ref = tr.get('gflop_per_launch')
            tbytes = tr.get('bytes_per_launch')
            if tbytes is not None and per_launch_ref:           # the PMC pass ran the same kernel: scale by the work of this run's launches
                tbytes = tbytes * (dom['flops'] / dom['launches'] / 1e9) / per_launch_ref
            all_ms = sum(v['ms'] for v in summ.values())
            all_fl = sum(v['flops'] for v in summ.values())
            launch_set = {}
            rgbk = (H.V2RGB_CONFIG, dom_id[1])
            if dom_id[0] == H.V2_CONFIG and rgbk in summ and summ[rgbk]['ms'] > 0:
                # the fourth 77 GFLOP launch of a step (SR block 1 conv1 forward) is another instantiation since round 5 -- it carries the toRGB layer in
                # its epilogue -- so `frac` above averages block 0 conv1 forward and the two DATA GRADIENTS (the slower direction); the four
                # launches together, as rounds 1-4 reported them:
                both_fl, both_ms = dom['flops'] + summ[rgbk]['flops'], dom['ms'] + summ[rgbk]['ms']
                launch_set = dict(launch_set='SR block 0 conv1 forward + the data gradients of both 77 GFLOP layers; block 1 conv1 forward runs as '
                                             'conv_v2_kernel<9,true,false,4,true,1> (same main loop + the 1x1 toRGB head in the epilogue: see `families`)',
                                  four_launch_tflops=round(both_fl / (both_ms * 1e-3) / 1e12, 2),
                                  four_launch_frac=round(both_fl / (both_ms * 1e-3) / 1e12 / (FP32_MFMA_PEAK_TFLOPS if dom_prec == 'f32' else BF16_MFMA_PEAK_TFLOPS), 4))
            # Every fraction below follows from a guide peak (MI355X_MICROARCH.md: dense 16-bit MFMA 2500 TFLOP/s, fp32 MFMA 157.3) and a
            # number measured in this run.  `frac` = ALGORITHMIC TFLOP/s / the peak of the instruction the kernel issues; a three-product
            # kernel executes 3 MFMA flops per algorithmic flop, so its executed fraction (`frac_executed`, what the MfmaUtil counter sees)
            # is three times that.
            hw_peak = FP32_MFMA_PEAK_TFLOPS if dom_prec == 'f32' else BF16_MFMA_PEAK_TFLOPS
            fams = []
            for k, v in summ.items():
                if v['ms'] <= 0:
                    continue
                pk = FP32_MFMA_PEAK_TFLOPS if {vv: kk for kk, vv in H.PRECISIONS.items()}[k[1]] == 'f32' else BF16_MFMA_PEAK_TFLOPS
                name = {H.V2_CONFIG: 'conv_v2<8 rows>', H.V2RGB_CONFIG: 'conv_v2<8 rows, 1x1 head> (SR block 1 conv1 forward + its toRGB)', H.V2H_CONFIG: 'conv_v2<4 rows>', H.V2Q_CONFIG: 'conv_v2<2 rows>', H.UP2_CONFIG: 'conv_v2_up2',
                        H.S2ADJ_CONFIG: 'conv_v2_s2adj', H.V3_CONFIG: 'conv_v3', H.WS_CONFIG: 'conv_ws'}.get(k[0], 'conv_igemm<%s>' % H.TILE_NAMES.get(k[0], '?'))
                fams.append(dict(kernel=name + ' / ' + {vv: kk for kk, vv in H.PRECISIONS.items()}[k[1]], launches_per_step=v['launches'] / args.steps,
                                 ms_per_step=round(v['ms'] / args.steps, 4), tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1),
                                 frac=round(v['flops'] / (v['ms'] * 1e-3) / 1e12 / pk, 4), share_of_conv_time=round(v['ms'] / all_ms, 3)))
            fams.sort(key=lambda f: -f['ms_per_step'])
            worst = min((f for f in fams if f['share_of_conv_time'] >= 0.05), key=lambda f: f['frac'], default=None)
            roof = dict(bound='mfma', kernel=kern, achieved=round(ach, 2),
                        peak=hw_peak, unit='TFLOP/s', frac=round(ach / hw_peak, 4), traffic=tbytes, traffic_source=tr.get('source'),
                        peak_basis=('fp32 matrix peak (MI355X_MICROARCH.md)' if dom_prec == 'f32' else 'dense 16-bit matrix peak of v_mfma_f32_32x32x16_f16 (MI355X_MICROARCH.md); achieved = algorithmic flops'),
                        products_per_fp32_product=nprod, mfma_executed_tflops=round(ach * nprod, 1), frac_executed=round(ach * nprod / hw_peak, 4),
                        mfma_register_loop_tflops_measured=round(probe_tf, 1) if probe_tf is not None else None,
                        frac_of_sustained=round(ach * nprod / probe_tf, 4) if probe_tf else None,       # executed TFLOP/s / what a register-only loop of the same instruction sustains on this part: the schedule headroom left
                        mfma_register_loop_note=('eg3d_probe_mfma_f16: register-only v_mfma_f32_32x32x16_f16 loop on RANDOM fp16 data, 1024 blocks x 4 waves x 8 accumulators, timed with HIP events '
                                                 'in this run right after the timed region.  The guide\'s 2495 TFLOP/s is the same instruction on its own benchmark; this probe reads 1.4-1.6 PFLOP/s on random data '
                                                 'and 2.0-2.3 on zeros on every box of this pool (tools/proto/mfma_peak.hip: the part clocks down under dense 16-bit MFMA load -- the guide\'s "DVFS give-back": '
                                                 'zero-filled inputs +19 % at equal cycle counts), i.e. it measures the sustained clock under this data, not a weak loop; executed (not algorithmic) TFLOP/s'),
                        frac_of_fp32_mfma_peak=round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                        launches_per_step=dom['launches'] / args.steps, gflop_per_launch=round(dom['flops'] / dom['launches'] / 1e9, 3),
                        avg_launch_ms=round(dom['ms'] / dom['launches'], 4), share_of_conv_time=round(dom['ms'] / all_ms, 3),
                        all_conv_tflops=round(all_fl / (all_ms * 1e-3) / 1e12, 1), all_conv_ms_per_step=round(all_ms / args.steps, 3),
                        all_conv_frac=round(all_fl / (all_ms * 1e-3) / 1e12 / hw_peak, 4),
                        all_conv_scope=('launches of the implicit-GEMM family (conv_igemm / conv_v2 / conv_v3 / up2 / s2adj); the low-latency toRGB launches of the '
                                        '4^2 .. 64^2 blocks (fp32 matrix pipe, 0.5 of 579 GFLOP per step) are not in it'),
                        families=fams, furthest_from_roofline=worst, **launch_set,
                        step_gflop_algorithmic=round(611.6 * M, 1), step_tflops=round(611.6e9 * M / (elapsed / args.steps) / 1e12, 1),
                        step_frac=round(611.6e9 * M / (elapsed / args.steps) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                        step_note='whole step: SURVEY 8d algorithmic 611.6 GFLOP per image-step (forward + data gradient, activation-scaled formulation) / ms_per_step of the timed region / 2500',
                        timing=roofline_pass)
        sp = prof.span_summary()
        if roof is not None:
            # backbone-only MFMA utilisation = 93.1 GF * k / (t_backbone * peak), k = conv passes executed (2: forward + data gradient)
            try:
                bf_ms, bb_ms = measure_backbone(G, dev, M)
                tb = (bf_ms + bb_ms) * 1e-3
                roof['roofline_backbone'] = dict(gflop_algorithmic=round(93.1 * 2 * M, 1), fwd_ms=round(bf_ms, 3), bwd_ms=round(bb_ms, 3),
                                                 tflops=round(93.1e9 * 2 * M / tb / 1e12, 1), peak=BF16_MFMA_PEAK_TFLOPS,
                                                 frac=round(93.1e9 * 2 * M / tb / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                                                 frac_executed=round(93.1e9 * 2 * 3 * M / tb / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                                                 note=('StyleGAN2 backbone only (4^2 .. 256^2: 90.1 GF of 3x3 convs + 3.0 GF toRGB per pass), forward and '
                                                       'backward (d planes -> d ws, d noise maps; weights frozen) each replayed from its own HIP graph and '
                                                       'timed with HIP events; every epilogue / FIR / style pass counts as time, none as flops; the '
                                                       'time-weighted MfmaUtil counter of the same kernels is in profiles/'))
            except Exception as e:         # a side measurement must never cost the benchmark line
                roof['roofline_backbone'] = dict(error='%s: %s' % (type(e).__name__, e))
        if 'render_fwd' in sp and 'render_bwd' in sp:
            # fused renderer forward 34.1 MB per image (planes 25.17 + rays 0.39 + uniforms 6.29 + outputs 2.23), backward
            # adds the 25.17 MB plane-gradient write and re-reads the forward's inputs
            alg = (34.1e6 + 34.1e6 + 25.17e6) * M
            t = (sp['render_fwd']['ms'] + sp['render_bwd']['ms']) / args.steps * 1e-3
            rt = traffic.get('renderer', {})
            roof_r = dict(bound='hbm', kernel='volume renderer: coarse_pos + (gather_rows + decode_rows) x2 + render<2>,<3> (forward); render<1> + decode_rows<true> + scatter_* (backward)',
                          achieved=round(alg / t / 1e9, 1), peak=8000.0, unit='GB/s', frac=round(alg / t / 1e9 / 8000.0, 4),
                          algorithmic_bytes_per_step=alg, fwd_ms=round(sp['render_fwd']['ms'] / args.steps, 3), bwd_ms=round(sp['render_bwd']['ms'] / args.steps, 3),
                          traffic=rt.get('bytes_per_step'), traffic_source=rt.get('source'), timing=roofline_pass)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_c2()
    cpu_c1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_c1 = cpu_baseline_c1()
    # The side figures and the full-budget run each capture a dozen graphs and build optimisers over all 30.7 M weights: they run in a process
    # of their own, AFTER this one has everything the benchmark line needs -- a fault in one of them (the runtime's graph capture has produced
    # segmentation faults under memory pressure) costs its own entry, never the line.
    side = final = None
    want_side = rank == 0 and world == 1 and not args.no_side_configs and M == 1 and args.loss_net == 'stub' and not args.wplus
    want_final = rank == 0 and world == 1 and not args.no_final_psnr
    if want_side or want_final:         # the children capture their own graphs: give the projector, its graphs and the optimiser state back first
        del proj
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    if want_side:
        side = run_child('side', args)
    if want_final:
        final = run_child('final', args)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        wl = ('C2: FFHQ 512^2 single-image latent inversion step' if M == 1 else
              f'C5 per-GPU share: {M} FFHQ 512^2 latent inversions as one batch (independent trajectories)')
        line = dict(metric='inversion-steps/sec (G fwd+bwd, 512^2 FFHQ EG3D) at 1/2/4/8 GPUs; final PSNR', value=round(world * M * args.steps / elapsed, 3),
                    unit='steps/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype={'f32': 'f32', 'bf16x6': 'f32 (bf16x6 split products, fp32-equivalent)',
                           'f16x3': 'f32 (modulated convs: two-piece fp16 split, 3 products, range-normalised; other GEMMs bf16x6; fp32-equivalent)',
                           'bf16x3': 'f32 storage, bf16x3 products (~2^-15)'}[prec_name], data='synthetic',
                    config=dict(workload=wl + ' (Phase A, w%s + 17 noise maps per image; G.synthesis fwd+bwd, 128^2 x 96-sample rendering, %s feature '
                                              'distance + noise regulariser, Adam)' % ('+' if args.wplus else '', 'stub-LPIPS' if args.loss_net == 'stub' else 'VGG16-LPIPS (256^2, random weights)'),
                                images_per_gpu=M, image_steps_per_timed_step=M, world_size=world, host_cpus_of_rank0=len(cpus),
                                generator='ffhqrebalanced512-128-shaped, 30.66 M params, random-init (synthetic weights)',
                                parallelism=f'{world * M} independent images, {M} per GPU; stat all-reduce only' + (
                                    ' (one packed vector per step, asynchronous; mean loss over ranks and steps %.5g)' % float(step_stats[0] / step_stats[2].clamp(min=1)) if world > 1 else ''),
                                launch='one HIP graph replay per step' if use_graph else 'eager (one launch per kernel)',
                                psnr_after_timed_steps_db=round(psnr_now, 3)),
                    spread=dict(repeats=len(regions), steps_per_region=args.steps, statistic='median region (value / ms_per_step); each region = --steps steps between two device synchronisations + barriers, max over ranks',
                                steps_per_s_min=round(world * M * args.steps / max(regions), 3), steps_per_s_median=round(world * M * args.steps / elapsed, 3),
                                steps_per_s_max=round(world * M * args.steps / min(regions), 3), ms_per_step_regions=[round(r / args.steps * 1e3, 4) for r in regions],
                                rel_spread=round((max(regions) - min(regions)) / elapsed, 4)),
                    roofline=roof, roofline_renderer=roof_r, cpu_baseline=cpu, cpu_baseline_c1=cpu_c1, side_configs=side, final_psnr=final)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
