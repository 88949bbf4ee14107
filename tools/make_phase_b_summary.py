"""profiles/<name>.md from a rocprofv3 kernel trace of `tools/time_phase_b.py graph`: per-kernel table of the last replayed pivotal-tuning step.
usage: python tools/make_phase_b_summary.py <dir with *_kernel_trace.csv> <out.md> ["Phase B step: ... line printed by the run"]"""
import csv, glob, re, sys, collections
src, out = sys.argv[1], sys.argv[2]
line = sys.argv[3] if len(sys.argv) > 3 else ''
unprof = sys.argv[4] if len(sys.argv) > 4 else ''          # the same script's line from an UN-profiled run in the same GPU call
tr = list(csv.DictReader(open(glob.glob(src + '/*kernel_trace.csv')[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'scatter_accum' in r['Kernel_Name']]           # (scatter_accum_kernel | scatter_accum16p_kernel | ...)
seg = tr[idx[-2] + 1:idx[-1] + 1]                     # one step = from after a tri-plane scatter to the next one (inclusive)
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
busy, span = sum(map(dur, seg)), int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)E', n)
    if m: return m.group(1)
    return re.sub(r'\(.*', '', n)[:100]
d = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = short(r['Kernel_Name']); d[k][0] += dur(r); d[k][1] += 1
with open(out, 'w') as f:
    f.write('# rocprofv3 kernel trace of the pivotal-tuning (Phase B, config C4) step replayed from a HIP graph\n\n')
    f.write('Command (GPU box, `cd /tmp; export TMPDIR=/tmp`): `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o b -- python tools/time_phase_b.py graph`\n')
    f.write('(full-size generator, all 30.7 M weights trainable, SR head in the reference\'s fp16-operand arithmetic, noise_mode=random, stub feature pyramid).\n\n')
    if line: f.write(f'Printed by the run: `{line}`\n\n')
    f.write(f'One replayed step: **{len(seg)} kernels, GPU-busy {busy / 1e6:.2f} ms, first-start to last-end {span / 1e6:.2f} ms** (profiler attached); '
            f'{sum(1 for r in seg if dur(r) < 8000)} of them run < 8 µs.\n\n')
    if unprof:
        f.write(f'Un-profiled, same GPU call: `{unprof}`.  The span above exceeds the busy time by {(span - busy) / 1e6:.2f} ms of gaps between kernels; the un-profiled step takes '
                f'about the BUSY time, so the gaps are the profiler\'s per-dispatch overhead on a {len(seg)}-node graph replay, not idle time of the product.\n\n')
    f.write('| kernel | launches | ms | avg µs |\n|---|---:|---:|---:|\n')
    for k, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0])[:40]:
        f.write(f'| `{k}` | {c} | {t / 1e6:.3f} | {t / c / 1e3:.1f} |\n')
print(len(seg), busy / 1e6, span / 1e6)
