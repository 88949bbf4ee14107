"""Per-image inversion driver: the caller of the hot path.

Counterpart of the reference's image loop (training/coaches/single_id_coach.py:27-84 on top of base_coach.py:52-99 and
training/projectors/w_projector.py:55-283), reduced to what touches the generator:

    for every image of this rank's shard:
        restore the pristine generator                        (restart_training: a fresh G per image, base_coach.py:52-61)
        Phase A  first_inv_steps of LatentProjector.step()    -> pivot latent (+ optimised camera)
        restore the generator's noise buffers                 (the reference projects on a deep copy of G, w_projector.py:68)
        Phase B  <= max_pti_steps of PivotalTuner.step(), early stop on the perceptual threshold (single_id_coach.py:69)
        metrics: MSE / PSNR of the pivot and of the tuned reconstruction
    one packed-stat all-reduce over the ranks at the end     (inv3d_amd.dist)

Images are independent optimisations, so N GPUs = N shards with no data-path communication (weak scaling).  File I/O, logging,
video / mesh export and the third-party encoders (e4e, ResNet pose head, ArcFace) of the reference loop are outside this package.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import os

import torch

from . import dist as D
from .inference import estimate_w_stats
from .inversion import LatentProjector, PivotalTuner, psnr_01


@dataclass
class InversionResult:
    name: str
    w_pivot: torch.Tensor                 # [1, num_ws, w_dim]
    cam: torch.Tensor                     # [1, 25]
    psnr_pivot: float
    psnr_tuned: float
    mse_tuned: float
    steps_a: int
    steps_b: int
    tuned_state: Optional[Dict[str, torch.Tensor]] = field(default=None, repr=False)    # generator weights after Phase B (opt-in)


class InversionCoach:
    def __init__(self, G, *, first_inv_steps: int = 400, max_pti_steps: int = 400, lpips_threshold: float = 0.06, first_inv_lr: float = 8e-3,
                 pti_lr: float = 3e-4, optimize_pose: bool = False, use_warping_loss: bool = False, wplus: bool = False,
                 feature_net: Optional[Callable] = None, early_stop_interval: int = 1, use_graph: bool = False, keep_tuned_state: bool = False,
                 synth_kwargs: Optional[dict] = None, seed: int = 0, pose_net_factory: Optional[Callable] = None, pose_mode: str = 'quat',
                 w_avg_samples: int = 10000, w_stats: Optional[Tuple[torch.Tensor, float]] = None,
                 start_w_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, sr_fp16: bool = True):
        """Hyper-parameter names and defaults follow configs/hyperparameters.py.  `early_stop_interval` = how often Phase B reads the
        early-stop state back to the host (1 = every step like the reference).  With the library's Adam the TEST itself runs on the device
        in every step whatever the interval (PivotalTuner.device_stop): the interval then only bounds how many masked no-op steps are
        issued after the stop, not when tuning stops.  Phase A starts where the reference starts (w_projector.py:88-97,100,118): at the mean latent
        of `w_avg_samples` mapped z (plus `start_w_fn(target_255_256)`, the e4e encoder's offset, when given) with the latent-noise scale
        tied to their standard deviation; `w_stats=(w_avg, w_std)` overrides the estimate, `w_avg_samples=0` starts at w = 0, std 1."""
        self.G = G
        self.first_inv_steps, self.max_pti_steps, self.thr = first_inv_steps, max_pti_steps, lpips_threshold
        self.first_inv_lr, self.pti_lr = first_inv_lr, pti_lr
        self.optimize_pose, self.use_warp, self.wplus = optimize_pose, use_warping_loss, wplus
        self.feature_net, self.interval, self.use_graph, self.keep = feature_net, max(1, early_stop_interval), use_graph, keep_tuned_state
        self.synth_kwargs, self.seed = dict(synth_kwargs or {}), seed
        self.pose_net_factory = pose_net_factory      # () -> a fresh pose estimator per image (the reference deep-copies its encoder, w_projector.py:62)
        self.pose_mode, self.start_w_fn = pose_mode, start_w_fn
        self.sr_fp16 = sr_fp16                        # Phase B runs the SR head like the reference does (PivotalTuner); Phase A is always fp32-equivalent
        if w_stats is not None:
            self.w_avg, self.w_std = w_stats[0], float(w_stats[1])
        elif w_avg_samples > 0:
            self.w_avg, self.w_std = estimate_w_stats(G, num_samples=w_avg_samples)     # once per generator: the weights are restored per image
        else:
            self.w_avg, self.w_std = None, 1.0
        # pristine copy of every parameter and buffer: what "re-loading the generator" means without a pickle on disk
        self._pristine = {k: v.detach().clone() for k, v in G.state_dict().items()}

    def restore_generator(self, only_buffers: bool = False):
        with torch.no_grad():
            bufs = {k for k, _ in self.G.named_buffers()}
            for k, v in self.G.state_dict().items():
                if only_buffers and k not in bufs:
                    continue
                v.copy_(self._pristine[k])
        for b in self.G.buffers():
            b.requires_grad = False
            b.grad = None

    def invert(self, name: str, target: torch.Tensor, cam: Optional[torch.Tensor] = None) -> InversionResult:
        G = self.G
        self.restore_generator()
        # ---- Phase A: latent (+ pose) with frozen weights ----------------------------------------------------------------------
        G.requires_grad_(False)
        start_w = None
        if self.start_w_fn is not None:
            with torch.no_grad():
                t255 = (target + 1) * (255 / 2)
                if t255.shape[2] > 256:
                    t255 = torch.nn.functional.interpolate(t255, size=(256, 256), mode='area')
                start_w = self.start_w_fn(t255)                       # e4e_enc(target_e4e).unsqueeze(1)  (w_projector.py:71-74,100)
        proj = LatentProjector(G, target, num_steps=self.first_inv_steps, cam=cam, optimize_pose=self.optimize_pose,
                               use_warping_loss=self.use_warp, first_inv_lr=self.first_inv_lr, wplus=self.wplus,
                               feature_net=self.feature_net, synth_kwargs=self.synth_kwargs, seed=self.seed,
                               use_graph=self.use_graph, w_avg=self.w_avg, w_std=self.w_std, start_w=start_w, pose_mode=self.pose_mode,
                               pose_net=self.pose_net_factory() if (self.pose_net_factory is not None and self.optimize_pose) else None)
        out = {}
        for _ in range(self.first_inv_steps):
            out = proj.step()
        w = proj.w_opt.detach()
        w_pivot = (w.repeat(1, G.backbone.num_ws, 1) if w.shape[1] == 1 else w).clone()
        cam_pivot = out['cam'].detach().clone() if out else proj.cam.detach().clone()
        self.restore_generator(only_buffers=True)
        with torch.no_grad():
            img = G.synthesis(w_pivot, cam_pivot, noise_mode='const', force_fp32=True, **self.synth_kwargs)['image']
            psnr_pivot = float(psnr_01(img, target))
        # ---- Phase B: generator weights around the pivot ------------------------------------------------------------------------
        tuner = PivotalTuner(G, target, w_pivot, cam_pivot, lr=self.pti_lr, lpips_threshold=self.thr, feature_net=self.feature_net,
                             synth_kwargs=self.synth_kwargs, sr_fp16=self.sr_fp16, use_graph=self.use_graph)
        steps_b = 0
        if tuner.hip_adam and tuner.device_stop:
            # the criterion is evaluated ON THE DEVICE in every step (a sticky flag that masks the update from the step at which it is met:
            # the reference's every-step `break` before the update, single_id_coach.py:68-71) -- every step can be a graph replay, and the
            # host only polls the flag (every `early_stop_interval` steps) to stop issuing work.  The number of updates made is the
            # optimiser's own device-side step count.
            for i in range(self.max_pti_steps):
                tuner.step()
                if (i % self.interval) == self.interval - 1 and tuner.stopped():
                    break
            steps_b = int(round(float(tuner.optimizer.step_t.item())))
        else:
            for i in range(self.max_pti_steps):
                check = (i % self.interval) == self.interval - 1
                res = tuner.step(early_stop=check)
                if check and res.get('done'):         # the reference leaves before the update (single_id_coach.py:68-71)
                    break
                steps_b += 1
        # how the two phases were actually issued (a refused capture falls back to eager launches with a warning: callers that quote timings check this)
        self.last_launch_modes = dict(phase_a='graph' if getattr(proj, '_graph', None) is not None else 'eager',
                                      phase_b='graph' if getattr(tuner, '_graph', None) is not None else 'eager')
        with torch.no_grad():
            img = G.synthesis(w_pivot, cam_pivot, noise_mode='const', force_fp32=True, **self.synth_kwargs)['image']
            psnr_tuned = float(psnr_01(img, target))
            mse = float(((img.clamp(-1, 1) - target) ** 2).mean() / 4.0)
        state = {k: v.detach().clone() for k, v in G.state_dict().items()} if self.keep else None
        G.requires_grad_(False)
        return InversionResult(name, w_pivot, cam_pivot, psnr_pivot, psnr_tuned, mse, self.first_inv_steps, steps_b, state)

    def run(self, images: Sequence[Tuple[str, torch.Tensor, Optional[torch.Tensor]]]) -> Tuple[List[InversionResult], Dict[str, float]]:
        """Invert this rank's shard of `images` = [(name, target [1,3,H,W] in [-1,1], cam [1,25] | None), ...] (every rank passes the
        same full list).  Returns (results of this rank, stats summed / maxed over all ranks)."""
        rank, world, local = D.init_from_env()
        if world > 1 and len(images):
            D.pin_to_numa(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
            D.warm_up(images[0][1].device)          # the communicator exists before any step is captured
        results = []
        for i in D.shard_images(len(images), rank, world):
            name, target, cam = images[i]
            results.append(self.invert(name, target, cam))
        dev = images[0][1].device if len(images) else torch.device('cpu')
        stats = D.allreduce_stats(dict(psnr=sum(r.psnr_tuned for r in results), mse=sum(r.mse_tuned for r in results),
                                       n_done=float(len(results)), steps=float(sum(r.steps_a + r.steps_b for r in results))), dev)
        if stats['n_done'] > 0:
            stats['mean_psnr'] = stats['psnr'] / stats['n_done']
        self.restore_generator()
        return results, stats
