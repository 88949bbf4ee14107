"""Per-kernel durations of one graph-replayed step from a rocprofv3 --kernel-trace CSV:  python tools/step_trace.py <dir> [min_us]"""
import csv, glob, re, sys, collections
d = sys.argv[1]; mn = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_apply_norm_kernel' in r['Kernel_Name'] or 'noise_apply_norm_kernel' in r['Kernel_Name']]
a, b = idx[-6], idx[-5]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]['Start_Timestamp'])
def short(n):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n)
    if m: return n[len(m.group(0)):][:int(m.group(1))]
    return re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    k = short(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    if (e - s) / 1e3 >= mn:
        print(f'{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  grid {int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])):>6}x{r["Grid_Size_Y"]}  {k}')
print('---- totals: kernels', len(seg), 'busy', round(sum(v[1] for v in agg.values()), 1), 'us; span', (int(seg[-1]['End_Timestamp']) - t0) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{v[0]:4d} {v[1]:8.1f}  {k}')
