"""Busy / idle time of the last steps of a bench.py run from `rocprofv3 --kernel-trace --output-format csv` (steps are delimited by the
noise renormalisation kernel); lists the largest idle gaps and which kernel follows them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_apply_norm_kernel' in r['Kernel_Name'] or 'noise_apply_norm_kernel' in r['Kernel_Name']]
for a, b in list(zip(idx[:-1], idx[1:]))[-4:-1]:
    seg = rows[a + 1:b + 1]
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
    union, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s <= ce: ce = max(ce, e)
        else: union += ce - cs; cs, ce = s, e
    union += ce - cs
    span = iv[-1][1] - int(rows[a]['End_Timestamp'])
    cur, gaps = int(rows[a]['End_Timestamp']), []
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > cur: gaps.append((s - cur, r['Kernel_Name'][:50]))
        cur = max(cur, e)
    gaps.sort(reverse=True)
    names = collections.Counter(r['Kernel_Name'][:40] for r in seg if 'rocclr' in r['Kernel_Name'])
    print(f'kernels {len(seg)}  busy {union/1e6:.3f} ms  span {span/1e6:.3f} ms  idle {(span-union)/1e6:.3f} ms  gaps>20us {[round(g[0]/1e3) for g in gaps if g[0] > 20000]}  blit nodes {dict(names)}')
