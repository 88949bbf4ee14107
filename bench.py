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
  roofline      : the dominant kernel conv_igemm_kernel<128,128,2,2,PREC> (implicit GEMM on the matrix cores): algorithmic FLOPs of
                  its launches (2 x MACs of the convolution) / their HIP-event durations measured on the
                  launch stream inside the timed region.  `peak` is the matrix peak for the arithmetic the kernel executes:
                  157.3 TFLOP/s for --precision f32 (v_mfma_f32_32x32x2_f32); for the split modes (fp32 operands and results, every
                  fp32 product formed from exact 16-bit MFMA products: 3 fp16 products in the default mode, 6 bf16 products with
                  --precision bf16x6) it is the dense 16-bit peak / products = 833.3 resp. 416.7 fp32-equivalent TFLOP/s.
                  `frac_of_fp32_mfma_peak` and `mfma_executed_tflops` are given beside it.
  cpu_baseline  : the CPU oracle (oracle/eg3d_oracle.py, a port of the reference's pure-PyTorch `_ref` path, pinned against
                  the reference) running the same C2 step on the host cores, bounded sample, rank 0 at N=1 only.
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
CONV_TRAFFIC_BYTES = 209.9e6       # HBM bytes per launch of the dominant kernel (both epilogue instantiations, launch-weighted): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, profiles/r01_bench_c2_summary.md
DOMINANT = 0                        # tile configuration id of conv_igemm_kernel<128,128,2,2>


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel from the host instead of replaying the captured step')
    ap.add_argument('--precision', default=None, choices=sorted(PRODUCTS) + ['auto'],
                    help="matrix-core arithmetic of the implicit GEMMs (default 'auto': f16x3 for the modulated convs, bf16x6 elsewhere)")
    ap.add_argument('--wplus', action='store_true')
    ap.add_argument('--loss-net', default='stub', choices=['stub', 'vgg16'],
                    help="feature network of the LPIPS term: 'stub' = the small fixed conv pyramid the C2 workload is defined with (SURVEY.md "
                         "section 8d); 'vgg16' = the full VGG16-LPIPS architecture (random weights) on the same kernels")
    args = ap.parse_args()

    from inv3d_amd import dist as D
    rank, world, local = D.init_from_env('nccl')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback in the product path)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from inv3d_amd import synthetic as S, hipops as H
    if args.precision is not None:
        H.set_conv_precision(args.precision)
    prec_name = H.modconv_precision()          # arithmetic of the dominant kernel's launches
    from inv3d_amd.inversion import LatentProjector, psnr_01

    G = S.make_generator(device=dev)
    S.load_synthetic_weights(G, seed=0)
    # per-rank independent image: target = render of a different latent by the same generator (so PSNR is meaningful)
    cam = S.synth_cameras(world, seed=2)[rank:rank + 1].to(dev)
    with torch.no_grad():
        ws_t = S.synth_ws(14, 512, world, seed=3)[rank:rank + 1].to(dev)
        target = G.synthesis(ws_t, cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
    use_graph = not args.no_graph
    feature_net = None
    if args.loss_net == 'vgg16':
        from inv3d_amd.loss_nets import VGG16LPIPS
        feature_net = VGG16LPIPS().to(dev)
    proj = LatentProjector(G, target, num_steps=400, cam=cam, wplus=args.wplus, seed=100 + rank, use_graph=use_graph, feature_net=feature_net)
    proj.preheat = 0

    stats = torch.zeros(4, device=dev)
    ones = torch.ones(1, device=dev)

    def one_step(pr=proj):
        out = pr.step()
        if world > 1:       # the path's only collective: packed per-step stats, summed on the device, never read inside the timed region
            stats.copy_(torch.stack([out['loss'], out['dist'], ones[0], ones[0]]))
            D.allreduce_stats_device(stats)
        return out

    if use_graph:                       # set-up, not a step of the benchmark: eager passes + the capture of the step into a HIP graph
        for _ in range(proj._graph_warmup + 1):
            one_step()
    for _ in range(args.warmup):
        one_step()
    prof = None
    if not args.no_roofline and not use_graph:
        prof = H.LaunchProfiler(only_config=DOMINANT)
        H.PROFILER = prof
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    H.PROFILER = None
    elapsed = D.max_over_ranks(elapsed, dev)
    final_psnr = float(psnr_01(proj.last['image'], target))
    roofline_pass = 'HIP events around every launch of the kernel inside the timed region'
    if not args.no_roofline and use_graph:
        # HIP events cannot bracket a kernel inside a captured graph: the per-launch durations of the dominant kernel come from an
        # instrumented EAGER pass over the same K steps (same generator, same shapes), run right after the timed region.
        roofline_pass = 'HIP events around every launch of the kernel in an eager re-run of the same %d steps right after the timed (graph-replay) region' % args.steps
        eager = LatentProjector(G, target, num_steps=400, cam=cam, wplus=args.wplus, seed=100 + rank, use_graph=False, feature_net=feature_net)
        eager.preheat = 0
        one_step(eager)
        prof = H.LaunchProfiler(only_config=DOMINANT)
        H.PROFILER = prof
        for _ in range(args.steps):
            one_step(eager)
        torch.cuda.synchronize()
        H.PROFILER = None

    roof = None
    if prof is not None:
        summ = prof.summary()
        dom = summ.get(DOMINANT)
        if dom and dom['ms'] > 0:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            nprod = PRODUCTS[prec_name]
            peak = FP32_MFMA_PEAK_TFLOPS if prec_name == 'f32' else BF16_MFMA_PEAK_TFLOPS / nprod
            kern = ('conv_igemm_kernel<128,128,2,2,0,*> (v_mfma_f32_32x32x2_f32)' if prec_name == 'f32' else
                    'conv_igemm_kernel<128,128,2,2,%d,*> (fp32 in/out, %d x v_mfma_f32_32x32x16_%s per fp32 product)' % (
                        H.PRECISIONS[prec_name], nprod, 'f16' if prec_name == 'f16x3' else 'bf16'))
            roof = dict(bound='mfma', kernel=kern, achieved=round(ach, 2),
                        peak=round(peak, 1), unit='TFLOP/s', frac=round(ach / peak, 4), traffic=CONV_TRAFFIC_BYTES,
                        peak_basis=('fp32 matrix peak' if prec_name == 'f32' else 'dense 16-bit matrix peak 2500 / %d products' % nprod),
                        frac_of_fp32_mfma_peak=round(ach / FP32_MFMA_PEAK_TFLOPS, 4), mfma_executed_tflops=round(ach * nprod, 1),
                        launches_per_step=dom['launches'] / args.steps, gflop_per_launch=round(dom['flops'] / dom['launches'] / 1e9, 3),
                        avg_launch_ms=round(dom['ms'] / dom['launches'], 4), timing=roofline_pass)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_c2()
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = dict(metric='inversion-steps/sec (G fwd+bwd, 512^2 FFHQ EG3D) at 1/2/4/8 GPUs; final PSNR', value=round(world * args.steps / elapsed, 3),
                    unit='steps/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype={'f32': 'f32', 'bf16x6': 'f32 (bf16x6 split products, fp32-equivalent)',
                           'f16x3': 'f32 (modulated convs: two-piece fp16 split, 3 products, range-normalised; other GEMMs bf16x6; fp32-equivalent)',
                           'bf16x3': 'f32 storage, bf16x3 products (~2^-15)'}[prec_name], data='synthetic',
                    config=dict(workload='C2: FFHQ 512^2 single-image latent inversion step (Phase A, w%s + 17 noise buffers; G.synthesis fwd+bwd, '
                                         '128^2 x 96-sample rendering, %s feature distance + noise regulariser, Adam)' % ('+' if args.wplus else '', 'stub-LPIPS' if args.loss_net == 'stub' else 'VGG16-LPIPS (256^2, random weights)'),
                                images_per_gpu=1, generator='ffhqrebalanced512-128-shaped, 30.66 M params, random-init (synthetic weights)',
                                parallelism=f'{world} independent images, 1 per GPU; stat all-reduce only',
                                launch='one HIP graph replay per step' if (use_graph and proj._graph is not None) else 'eager (one launch per kernel)',
                                psnr_after_timed_steps_db=round(final_psnr, 3)),
                    roofline=roof, cpu_baseline=cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
