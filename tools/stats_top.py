"""Top kernels of a rocprofv3 --kernel-trace --stats run by total time:  python tools/stats_top.py <dir> [n]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = re.sub(r'\(anonymous namespace\)::|void ', '', r['Name'])[:80]
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}%  {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:8.1f}  {n}")
print(f'total {tot / 1e6:.2f} ms over {sum(int(r["Calls"]) for r in rows)} launches')
