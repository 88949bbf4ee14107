cd /root/repo
for v in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && EG3D_CONV_WS_S2=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/abs$v -o k -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side --no-final-psnr --no-roofline > /dev/null 2>&1)
  python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob('/root/repo/gpurun_out/abs$v/*kernel_trace.csv')[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_apply_norm' in r['Kernel_Name']]
seg=rows[idx[-2]+1:idx[-1]+1]
print('S2=$v kernels',len(seg),'busy us',sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e3)
for i,r in enumerate(seg):
    n=r['Kernel_Name']
    if i>100 and ('conv_ws' in n or 'Li32ELi128ELi1ELi4' in n or 'conv_igemm_kernel<32, 128, 1, 4' in n or 'fir44(' in n or 'dgrad_finish' in n):
        print('   ',i, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, n[:80])
PY
done
