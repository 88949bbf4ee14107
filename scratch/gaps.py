import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'noise_normalize_kernel' in r['Kernel_Name']]
print('steps found', len(idx))
for a,b in list(zip(idx[:-1],idx[1:]))[-4:]:
    seg=rows[a+1:b+1]
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)
    span=int(seg[-1]['End_Timestamp'])-int(rows[a]['End_Timestamp'])
    gaps=[int(y['Start_Timestamp'])-int(x['End_Timestamp']) for x,y in zip([rows[a]]+seg[:-1],seg)]
    h=collections.Counter(min(int(g/1000),10) for g in gaps)
    print(f'kernels {len(seg)} busy {busy/1e6:.3f} ms span {span/1e6:.3f} ms idle {(span-busy)/1e6:.3f}  gap hist(us) {sorted(h.items())}')
seg=rows[idx[-2]+1:idx[-1]+1]
gl=sorted(((int(y['Start_Timestamp'])-int(x['End_Timestamp']),x['Kernel_Name'][:45],y['Kernel_Name'][:45]) for x,y in zip(seg[:-1],seg[1:])),reverse=True)[:8]
for g in gl: print(g)
