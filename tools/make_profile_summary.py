"""Turn the outputs of one profiling session into profiles/<name>_summary.md (+ the --stats CSV next to it).

    python tools/make_profile_summary.py <bench.json> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix>

bench.json   : the JSON line of an un-profiled `python bench.py` on the same box
stats_dir    : rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -o b -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline
pmc_*_dir    : rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -d <dir> -o f|w -- python bench.py --steps 3 --warmup 1 ...
"""
import collections
import csv
import glob
import json
import shutil
import sys

bench_json, stats_dir, pf_dir, pw_dir, prefix = sys.argv[1:6]
bench = json.loads(open(bench_json).read())
rows = list(csv.DictReader(open(glob.glob(stats_dir + '/*_kernel_stats.csv')[0])))
ncalls = sum(int(r['Calls']) for r in rows)
tr = list(csv.DictReader(open(glob.glob(stats_dir + '/*_kernel_trace.csv')[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'noise_apply_norm_kernel' in r['Kernel_Name']]
a, b = idx[8], idx[9]                       # a graph-replayed step of the timed region (3 set-up + 2 warm-up steps precede it)
seg = tr[a + 1:b + 1]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    agg[r['Kernel_Name']][0] += 1
    agg[r['Kernel_Name']][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
busy = sum(v[1] for v in agg.values())
span = int(seg[-1]['End_Timestamp']) - int(tr[a]['End_Timestamp'])


def pmc(d, name):
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(d + '/*_counter_collection.csv')[0])):
        if r['Counter_Name'] == name:
            out[r['Kernel_Name']][0] += 1
            out[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return out


f, w = pmc(pf_dir, 'FETCH_SIZE'), pmc(pw_dir, 'WRITE_SIZE')
o = ['# rocprofv3 summary of `bench.py` (C2, N=1, default bf16x6 arithmetic, HIP-graph replay)\n',
     'Commands (GPU box, `cd /tmp; export TMPDIR=/tmp`):\n',
     '```\nrocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o b -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline\n'
     'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph\n'
     'rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph\n```\n',
     f'Un-profiled default `python bench.py` on the same box/call: **{bench["value"]} steps/s** ({bench["ms_per_step"]} ms/step).\n',
     f'`roofline`: `{json.dumps(bench["roofline"])}`\n', f'`cpu_baseline`: `{json.dumps(bench["cpu_baseline"])}`\n',
     f'One graph-replayed step in the kernel trace: {len(seg)} kernels, GPU-busy {busy/1e6:.2f} ms, first-start to last-end {span/1e6:.2f} ms '
     '(profiler attached; the noise regulariser overlaps the backbone on a graph branch).\n',
     '| kernel (one replayed step) | launches | ms | avg µs |\n|---|---:|---:|---:|']
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
    o.append(f'| `{k[:100]}` | {v[0]} | {v[1]/1e6:.3f} | {v[1]/v[0]/1e3:.1f} |')
dom = next(r for r in rows if 'conv_igemm_kernel<128, 128, 2, 2,' in r['Name'] and 'true>' in r['Name'])
o.append(f'\nWhole-run `--stats` table ({ncalls} launches incl. set-up, warm-up and the eager roofline pass): `{prefix.split("/")[-1]}_kernel_stats.csv`. '
         f'Dominant kernel there: `{dom["Name"][:90]}` {dom["Calls"]} calls, average {float(dom["AverageNs"])/1e3:.1f} µs '
         f'(bench.py HIP events, un-profiled: {bench["roofline"]["avg_launch_ms"]*1e3:.1f} µs).\n')
o.append('## HBM traffic (PMC, separate passes; FETCH_SIZE / WRITE_SIZE are reported in KiB)\n')
o.append('`FETCH_SIZE` on gfx950 counts 64 B per 128-B request for wide coalesced reads, so it is doubled below as MI355X_MICROARCH.md (HBM section) '
         'prescribes; `WRITE_SIZE` is used as reported (uncalibrated).\n')
o.append('| kernel | launches | FETCH_SIZE/launch (raw MB) | corrected read MB | WRITE_SIZE/launch MB | total MB/launch |\n|---|---:|---:|---:|---:|---:|')
for k in sorted(f, key=lambda k: -f[k][1])[:10]:
    fr = f[k][1] / f[k][0] / 1024
    wr = w[k][1] / max(w[k][0], 1) / 1024
    o.append(f'| `{k[:90]}` | {f[k][0]} | {fr:.2f} | {2*fr:.2f} | {wr:.2f} | {2*fr+wr:.2f} |')
open(prefix + '_summary.md', 'w').write('\n'.join(o) + '\n')
shutil.copy(glob.glob(stats_dir + '/*_kernel_stats.csv')[0], prefix + '_kernel_stats.csv')
dk = [k for k in f if 'conv_igemm_kernel<128, 128, 2, 2, 3, true>' in k]
if dk:
    k = dk[0]
    print('dominant kernel traffic MB/launch:', 2 * f[k][1] / f[k][0] / 1024 + w[k][1] / max(w[k][0], 1) / 1024)
print(len(seg), busy / 1e6, span / 1e6)
