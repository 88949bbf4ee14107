"""Per-kernel timeline of one graph-replayed Phase-B step from a rocprofv3 --kernel-trace CSV (steps are delimited by the fused Adam launches):
    python tools/phase_b_timeline.py <dir> > timeline.txt"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'FusedOptimizer' in r['Kernel_Name']]
per = 4                                   # Adam launches per step
a, b = idx[-2 * per - 1], idx[-per - 1]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])[:70]
    g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    print(f'{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} grid {g:>6}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} {n}')
print('kernels', len(seg), 'span', (int(seg[-1]['End_Timestamp']) - t0) / 1e3, 'us')
