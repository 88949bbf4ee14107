cd /root/repo
for v in 0 1; do
  (cd /tmp && export TMPDIR=/tmp && EG3D_RGB_HEAD=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/abk$v -o k -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side --no-final-psnr --no-roofline > /dev/null 2>&1)
  python - <<PY
import csv,glob,collections
rows=list(csv.DictReader(open(glob.glob('/root/repo/gpurun_out/abk$v/*kernel_trace.csv')[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_apply_norm' in r['Kernel_Name']]
seg=rows[idx[-2]+1:idx[-1]+1]
print('RGB_HEAD=$v kernels',len(seg),'busy us',sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e3)
for r in seg:
    n=r['Kernel_Name']
    if 'conv_v2_kernel' in n and ('ELi4E' in n or ', 4' in n) or 'Li128ELi32ELi4ELi1ELi3' in n or 'conv_igemm_kernel<128, 32, 4, 1, 3' in n:
        print('   ', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, n[:90])
PY
done
