"""Per-kernel totals of one graph-replayed pivotal-tuning step (rocprofv3 --kernel-trace CSV):  python tools/phase_b_step_totals.py <dir>"""
import csv, glob, re, sys, collections
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r['Kernel_Name']]
gaps = [idx[j + 1] - idx[j] for j in range(len(idx) - 1)]
per = 1
while per < len(gaps) and gaps[-per] == 1:
    per += 1
ends = idx[::-1][::per][::-1]
a, b = ends[-4], ends[-3]
seg = rows[a + 1:b + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
def short(n):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n)
    if m: return n[len(m.group(0)):][:int(m.group(1))]
    return re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
for r in seg:
    k = short(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('kernels', len(seg), 'busy', round(sum(v[1] for v in agg.values()), 1), 'us; span', (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{v[0]:4d} {v[1]:8.1f}  {k}')
