"""Two runs of an optimisation loop from the same state, compared bit for bit (run with EG3D_DETERMINISTIC=1 for the deterministic build;
with the normal build the runs differ after the first step).  Usage: det_runs.py [c2|c3|phase_b] [steps] [graph|eager]
Prints one JSON line: {"config", "steps", "graph", "deterministic_build", "equal", "max_abs_diff", "misses", "ms_per_step"}."""
import sys, json, time, copy
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), '3dgan-inversion_amd'))
import torch
from inv3d_amd import synthetic as S, _lib as L
from inv3d_amd.inversion import LatentProjector, PivotalTuner
which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
graph = (sys.argv[3] if len(sys.argv) > 3 else 'graph') == 'graph'
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
state0 = copy.deepcopy(G.state_dict())
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)

def run():
    G.load_state_dict(state0)
    torch.manual_seed(123)
    if which == 'phase_b':
        for p in G.parameters(): p.requires_grad_(True)
        T = PivotalTuner(G, target, S.synth_ws(14, 512, 1, seed=5).to(dev), cam, use_graph=graph)
        t0 = None
        for i in range(steps):
            if i == min(10, steps - 1): torch.cuda.synchronize(); t0 = time.perf_counter(); i0 = i
            T.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / max(1, steps - i0) * 1e3
        return [p.detach().clone() for p in G.parameters()], ms
    P = LatentProjector(G, target, num_steps=max(steps, 20), optimize_pose=which == 'c3', use_warping_loss=which == 'c3', cam_preheat_steps=2, seed=1,
                        use_graph=graph)
    t0 = None
    for i in range(steps):
        if i == min(10, steps - 1): torch.cuda.synchronize(); t0 = time.perf_counter(); i0 = i
        P.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(1, steps - i0) * 1e3
    out = [P.w_opt.detach().clone()] + [b.detach().clone() for b in P._all_bufs]
    if which == 'c3':
        out += [p.detach().clone() for g in P.cam_optimizer.param_groups for p in g['params']]
        out += [p.detach().clone() for g in P.translation_optimizer.param_groups for p in g['params']]
    return out, ms

run()                                  # (captures, caches and the arena reach their steady state)
a, ms = run(); b, _ = run()
diff = max(float((x.double() - y.double()).abs().max()) for x, y in zip(a, b))
print(json.dumps(dict(config=which, steps=steps, graph=graph, deterministic_build=bool(L.lib().eg3d_det_enabled()),
                      equal=all(torch.equal(x, y) for x, y in zip(a, b)), max_abs_diff=diff, misses=L.det_misses(), ms_per_step=round(ms, 3))))
