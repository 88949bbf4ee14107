"""Weight-gradient launches of one graph-replayed pivotal-tuning step from a rocprofv3 --kernel-trace CSV:  python tools/phase_b_wgrad_table.py <dir> [pattern ...]"""
import csv, glob, re, sys
d = sys.argv[1]
pats = sys.argv[2:] or ['wgrad', 'weight_grad', 'rows_gram']
rows = list(csv.DictReader(open(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r['Kernel_Name']]
gaps = [idx[j + 1] - idx[j] for j in range(len(idx) - 1)]
per = 1
while per < len(gaps) and gaps[-per] == 1:      # consecutive adam launches of one optimiser step
    per += 1
ends = idx[per - 1::per] if (len(idx) % per == 0) else idx[::-1][::per][::-1]
a, b = ends[-4], ends[-3]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0.0
for r in seg:
    n = r['Kernel_Name']
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if any(p in n for p in pats):
        short = re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {dur:7.1f} grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):>5}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {short}")
        tot += dur
print('total of the listed launches', round(tot, 1), 'us; step span', (int(seg[-1]['End_Timestamp']) - t0) / 1e3, 'us;', len(seg), 'kernels')
