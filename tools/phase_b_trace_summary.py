import csv,sys,collections,re
tr=list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in tr]
idx=[i for i,n in enumerate(names) if 'scatter_accum_kernel' in n]
a,b=idx[-2],idx[-1]
seg=tr[a+1:b+1]
print(len(seg), 'launches; busy ms', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e6, 'span ms', (int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6)
