"""Compare two rocprofv3 --kernel-trace --stats runs of the same command (A/B of an environment switch): per kernel, calls and total time.
    python tools/ab_kernel_stats.py <dirA> <dirB> [min_us]"""
import csv, glob, re, sys, collections


def short(name):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)
    if m:
        return name[len(m.group(0)):][:int(m.group(1))]
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name).replace(' ', '')[:90]


def load(d):
    f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r['Name'])
        out[k][0] += int(r['Calls'])
        out[k][1] += float(r['TotalDurationNs']) / 1e3
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
mn = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, [0, 0.0])
    cb, tb = b.get(k, [0, 0.0])
    if abs(tb - ta) >= mn:
        rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort()
print(f'{"delta us":>10} {"calls A":>8} {"us A":>10} {"calls B":>8} {"us B":>10}  kernel')
for d, k, ca, ta, cb, tb in rows:
    print(f'{d:10.0f} {ca:8d} {ta:10.0f} {cb:8d} {tb:10.0f}  {k}')
print(f'total A {sum(v[1] for v in a.values()) / 1e3:.2f} ms, B {sum(v[1] for v in b.values()) / 1e3:.2f} ms')
