"""Mean per-launch counter values per kernel from the rocprofv3 --pmc passes of tools/pmc_renderer.sh:  python tools/pmc_summary.py <dir> <kernel substring> ..."""
import csv, glob, sys, collections, re
d = sys.argv[1]; pats = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if any(p in k for p in pats):
            m = re.search(r'(\w+_kernel(<[^>]*>)?)', k)
            agg[m.group(1) if m else k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(agg.items()):
    print('==', k)
    for n, v in sorted(c.items()):
        print(f'   {n:40s} {sum(v) / len(v):16.1f}   ({len(v)} launches)')
