"""Run-to-run differences of the C2 step, layer by layer: the same eager step twice from the same state; every layer's output and the gradient
arriving at it are compared bit for bit (the first differing entry of each list names the kernel that is not order-fixed)."""
import os, sys
sys.path.insert(0, '/root/repo/3dgan-inversion_amd')
import torch
from inv3d_amd import synthetic as S
from inv3d_amd.inversion import LatentProjector
dev = torch.device('cuda')
G = S.make_generator(device=dev); S.load_synthetic_weights(G, seed=0)
cam = S.synth_cameras(1, seed=2).to(dev)
with torch.no_grad():
    target = G.synthesis(S.synth_ws(14, 512, 1, seed=3).to(dev), cam, noise_mode='const', force_fp32=True)['image'].clamp(-1, 1)
pose = '--pose' in sys.argv

def run(steps):
    torch.manual_seed(123)
    P = LatentProjector(G, target, num_steps=400, optimize_pose=pose, use_warping_loss=pose, cam_preheat_steps=1, seed=1, use_graph=False)
    fw, bw = {}, {}
    trail = []
    hooks = []
    for name, m in G.named_modules():
        if len(list(m.children())) == 0 or type(m).__name__ in ('SynthesisLayer', 'ToRGBLayer', 'SynthesisBlock', 'ImportanceRenderer', 'OSGDecoder'):
            def fh(mod, inp, out, name=name):
                outs = out if isinstance(out, (tuple, list)) else (out.values() if isinstance(out, dict) else (out,))
                for i, o in enumerate(outs):
                    if torch.is_tensor(o) and o.is_floating_point():
                        fw.setdefault(f'{name}.{i}', []).append(o.detach().clone())
                        if o.requires_grad:
                            o.register_hook(lambda g, key=f'{name}.{i}': bw.setdefault(key, []).append(g.detach().clone()) if g is not None else None)
            hooks.append(m.register_forward_hook(fh))
    for s in range(steps):
        fw.clear(); bw.clear()
        P.step()
        snap = {'w_opt': P.w_opt.detach().clone(), 'w_opt.grad': P.w_opt.grad.detach().clone() if P.w_opt.grad is not None else torch.zeros(1)}
        for k, v in list(P.noise_bufs.items()) + list(P.noise_bufs2.items()):
            snap['noise.' + k] = v.detach().clone()
            if v.grad is not None: snap['grad.' + k] = v.grad.detach().clone()
        snap['loss'] = torch.as_tensor(P.last['dist']).detach().clone() if isinstance(P.last, dict) and 'dist' in P.last else torch.zeros(1)
        trail.append((snap, {k: [t for t in v] for k, v in fw.items()}, {k: [t for t in v] for k, v in bw.items()}))
    for h in hooks: h.remove()
    leaves = {'w_opt': P.w_opt.detach().clone()}
    for k, v in P.noise_bufs.items(): leaves['noise.' + k] = v.detach().clone()
    if pose: leaves['pose'] = P.pose_vec.detach().clone() if hasattr(P, 'pose_vec') else torch.zeros(1)
    return fw, bw, leaves, trail

steps = int(os.environ.get('STEPS', '3'))
run(1); a = run(steps); b = run(steps)
def cmp(title, x, y):
    bad = 0
    for k in x:
        if k not in y: continue
        for i, (u, v) in enumerate(zip(x[k], y[k])):
            if not torch.equal(u, v):
                d = (u.double() - v.double()).abs().max().item(); m = u.double().abs().max().item()
                print(f'  {title} {k}[{i}] shape {tuple(u.shape)} max|diff| {d:.3e} (max|x| {m:.3e}), {(u != v).sum().item()} of {u.numel()} differ'); bad += 1
    print(f'{title}: {bad} differing of {sum(len(v) for v in x.values())}')
cmp('fwd', a[0], b[0]); cmp('bwd', a[1], b[1])
for k in a[2]:
    print('leaf', k, 'equal' if torch.equal(a[2][k], b[2][k]) else f'DIFF {(a[2][k].double() - b[2][k].double()).abs().max().item():.3e}')

for st, (ta, tb) in enumerate(zip(a[3], b[3])):
    bad = [k for k in ta[0] if not torch.equal(ta[0][k], tb[0][k])]
    print('step', st, 'differing leaves/grads:', bad[:12], '...' if len(bad) > 12 else '')
    if bad:
        cmp(f'step{st} fwd', ta[1], tb[1]); cmp(f'step{st} bwd', ta[2], tb[2])
        break
#print('fwd keys:', sorted(a[3][0][1].keys()))
#print('bwd keys:', sorted(a[3][0][2].keys()))
from inv3d_amd import _lib as L
print('deterministic build:', bool(L.lib().eg3d_det_enabled()), ' misses:', L.det_misses())
