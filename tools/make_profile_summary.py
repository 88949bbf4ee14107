"""Turn the outputs of one profiling session (tools/profile_session.sh -> gpurun_out/sess/) into profiles/<name>_summary.md, the --stats
CSV next to it, and profiles/traffic_table.json (per-kernel HBM bytes per launch from the PMC passes: what bench.py's `roofline.traffic`
is read from).

    python tools/make_profile_summary.py <session_dir> <out_prefix> ["title"]
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

sess, prefix = sys.argv[1:3]
title = sys.argv[3] if len(sys.argv) > 3 else 'C2, N=1, default f16x3 arithmetic, HIP-graph replay'
bench = json.loads(open(sess + '/bench.json').read().strip().splitlines()[-1])
cmds = [l for l in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profile_session.sh')) if 'rocprofv3' in l and l.startswith('timeout')]


def one(pattern):
    g = glob.glob(pattern)
    return g[0] if g else None


def short(name):
    """conv_igemm_kernel<128, 128, 2, 2, 3, true>(eg3d_conv_params) -> conv_igemm_kernel<128,128,2,2,3,true>"""
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)          # a symbol rocprofv3 left mangled (vector-of-_Float16 parameters)
    if m:
        return name[len(m.group(0)):][:int(m.group(1))]
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name).replace(' ', '')[:110]


def table_key(s):
    m = re.match(r'conv_igemm_kernel<(\d+,\d+,\d+,\d+),', s)
    if m:
        return 'conv_igemm_kernel<%s>' % m.group(1)
    return s


rows = list(csv.DictReader(open(one(sess + '/stats/*_kernel_stats.csv'))))
ncalls = sum(int(r['Calls']) for r in rows)
tr = list(csv.DictReader(open(one(sess + '/stats/*_kernel_trace.csv'))))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'adam_apply_norm_kernel' in r['Kernel_Name'] or 'noise_apply_norm_kernel' in r['Kernel_Name']]      # last launch of a step
idx = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 12]         # (a batch has several of them in a row: keep the last of each step)
a, b = idx[8], idx[9]                       # a graph-replayed step of the timed region (3 set-up + 2 warm-up steps precede it)
seg = tr[a + 1:b + 1]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = short(r['Kernel_Name'])
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
busy = sum(v[1] for v in agg.values())
span = int(seg[-1]['End_Timestamp']) - int(tr[a]['End_Timestamp'])


def pmc(d):
    """{counter: {kernel: [launches, sum]}} of one PMC pass (None when the pass did not produce a table)."""
    f = one(d + '/*_counter_collection.csv')
    if f is None:
        return None
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        e = out[r['Counter_Name']][short(r['Kernel_Name'])]
        e[0] += 1
        e[1] += float(r['Counter_Value'])
    return out


pf, pw, pm, pl = pmc(sess + '/pf'), pmc(sess + '/pw'), pmc(sess + '/pm'), pmc(sess + '/pl')
roof = bench.get('roofline') or {}
o = [f'# rocprofv3 summary of `bench.py` ({title})\n',
     'Commands (GPU box, `cd /tmp; export TMPDIR=/tmp`; tools/profile_session.sh):\n',
     '```\n' + ''.join(re.sub(r'timeout \d+ ', '', c).replace('$R/', '').replace('$O/', '<dir>/').split(' > ')[0] + '\n' for c in cmds) + '```\n',
     f'Un-profiled default `python bench.py` in the same call: **{bench["value"]} {bench["unit"]}** ({bench["ms_per_step"]} ms/step), '
     f'`final_psnr`: `{json.dumps(bench.get("final_psnr"))}`\n',
     f'`roofline`: `{json.dumps(roof)}`\n', f'`roofline_renderer`: `{json.dumps(bench.get("roofline_renderer"))}`\n',
     f'`cpu_baseline`: `{json.dumps(bench.get("cpu_baseline"))}`\n',
     f'One graph-replayed step in the kernel trace: {len(seg)} kernels, GPU-busy {busy/1e6:.2f} ms, first-start to last-end {span/1e6:.2f} ms '
     '(profiler attached).\n',
     '| kernel (one replayed step) | launches | ms | avg µs |\n|---|---:|---:|---:|']
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    o.append(f'| `{k}` | {v[0]} | {v[1]/1e6:.3f} | {v[1]/v[0]/1e3:.1f} |')
with open(prefix + '_step_kernels.csv', 'w') as fh:          # every launch of that step, in order (what a replay actually issues)
    fh.write('index,start_us,duration_us,grid,kernel\n')
    t0 = int(seg[0]['Start_Timestamp'])
    for i, r in enumerate(seg):
        fh.write('%d,%.1f,%.1f,%s,"%s"\n' % (i, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                             r.get('Grid_Size', ''), short(r['Kernel_Name'])))
o.append(f'\nAll {len(seg)} launches of that step in issue order: `{os.path.basename(prefix)}_step_kernels.csv`.\n')
dom_short = max((k for k in agg if k.startswith('conv_')), key=lambda k: agg[k][1])
dom = next(r for r in rows if short(r['Name']) == dom_short)
o.append(f'\nWhole-run `--stats` table ({ncalls} launches incl. set-up, warm-up and the eager roofline pass): `{os.path.basename(prefix)}_kernel_stats.csv`. '
         f'Dominant kernel there: `{dom_short}` {dom["Calls"]} calls, average {float(dom["AverageNs"])/1e3:.1f} µs '
         f'(bench.py HIP events, un-profiled: {roof.get("avg_launch_ms", 0)*1e3:.1f} µs).\n')

table = {}
if pf and pw:
    f, w = pf['FETCH_SIZE'], pw['WRITE_SIZE']
    o.append('## HBM traffic (PMC, separate passes; FETCH_SIZE / WRITE_SIZE are reported in KiB)\n')
    o.append('`FETCH_SIZE` on gfx950 counts 64 B per 128-B request for wide coalesced reads, so it is doubled below as MI355X_MICROARCH.md (HBM section) '
             'prescribes; `WRITE_SIZE` is used as reported (uncalibrated).\n')
    o.append('| kernel | launches | FETCH_SIZE/launch (raw MB) | corrected read MB | WRITE_SIZE/launch MB | total MB/launch | avg µs (same launches, FETCH pass) | TB/s |\n|---|---:|---:|---:|---:|---:|---:|---:|')
    # durations of the SAME launches the FETCH_SIZE counters belong to (the pass's own kernel trace; counters attached)
    dur = collections.defaultdict(lambda: [0, 0.0])
    tf = one(sess + '/pf/*_kernel_trace.csv')
    if tf:
        for r in csv.DictReader(open(tf)):
            e = dur[short(r['Kernel_Name'])]
            e[0] += 1
            e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    avg_us = {k: v[1] / v[0] for k, v in dur.items() if v[0]}
    per_launch = {}
    for k in f:
        fr = f[k][1] / f[k][0] / 1024
        wr = w[k][1] / max(w[k][0], 1) / 1024 if k in w else 0.0
        per_launch[k] = (f[k][0], fr, wr, (2 * fr + wr) * 1e6)
    for k in sorted(per_launch, key=lambda k: -per_launch[k][0] * per_launch[k][3])[:22]:
        n, fr, wr, tot = per_launch[k]
        us = avg_us.get(k)
        o.append(f'| `{k}` | {n} | {fr:.2f} | {2*fr:.2f} | {wr:.2f} | {tot/1e6:.2f} | ' + (f'{us:.1f} | {tot / us / 1e6:.2f} |' if us else '| |'))
    src = f'{os.path.basename(prefix)}_summary.md (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)'
    for k, (n, fr, wr, tot) in per_launch.items():
        if k.startswith('conv_'):
            e = table.setdefault(table_key(k), dict(bytes_per_launch=0.0, launches=0, source=src))
            e['bytes_per_launch'] = (e['bytes_per_launch'] * e['launches'] + tot * n) / (e['launches'] + n)
            e['launches'] += n
    dk = table_key(dom_short)
    if dk in table and roof.get('gflop_per_launch'):
        table[dk]['gflop_per_launch'] = roof['gflop_per_launch']
    ren = [k for k in agg if re.match(r'(render_kernel|decode_rows_kernel|gather_rows_kernel|coarse_pos_kernel|scatter_)', k)]
    rb = sum(agg[k][0] * per_launch[k][3] for k in ren if k in per_launch)
    table['renderer'] = dict(bytes_per_step=rb, kernels=sorted(ren), source=src)
    o.append(f'\nRenderer kernels of one step ({", ".join("`%s`" % k for k in sorted(ren))}): **{rb/1e6:.1f} MB** of HBM traffic per step.\n')
    json.dump(table, open(os.path.join(os.path.dirname(prefix), 'traffic_table.json'), 'w'), indent=1, sort_keys=True)

if pm:
    o.append('## Matrix-pipe and LDS counters (rocprofv3 derived metrics, averaged over the launches of each kernel in the PMC pass)\n')
    o.append('`MfmaUtil` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) x 100; `LdsUtil` = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE x CUs) x 100; '
             '`LdsBankConflict` = conflict cycles / conflict-free active cycles; GFLOP = SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 (executed 16-bit matrix work).\n')
    o.append('| kernel | launches | MfmaUtil % | LdsUtil % | LdsBankConflict | executed F16 MFMA GFLOP / launch |\n|---|---:|---:|---:|---:|---:|')
    mu, lu = pm.get('MfmaUtil', {}), pm.get('LdsUtil', {})
    bc, mo = (pl or {}).get('LdsBankConflict', {}), (pl or {}).get('SQ_INSTS_VALU_MFMA_MOPS_F16', {})
    avg = lambda t, k: t[k][1] / t[k][0] if k in t and t[k][0] else float('nan')
    for k in sorted((k for k in mu if k in agg), key=lambda k: -agg[k][1])[:16]:
        o.append(f'| `{k}` | {mu[k][0]} | {avg(mu, k):.1f} | {avg(lu, k):.1f} | {avg(bc, k):.3f} | {avg(mo, k)*512/1e9:.2f} |')
    conv = [k for k in mu if k.startswith('conv_') and k in agg]
    wsum = sum(agg[k][1] for k in conv)
    if wsum:
        o.append(f'\nAll conv kernels of one step, weighted by their time in the step: MfmaUtil **{sum(avg(mu, k) * agg[k][1] for k in conv) / wsum:.1f} %** '
                 f'over {wsum/1e6:.2f} ms ({len(conv)} kernel variants).\n')
open(prefix + '_summary.md', 'w').write('\n'.join(o) + '\n')
shutil.copy(one(sess + '/stats/*_kernel_stats.csv'), prefix + '_kernel_stats.csv')
print(len(seg), busy / 1e6, span / 1e6, dom_short, json.dumps(table.get(table_key(dom_short))))
