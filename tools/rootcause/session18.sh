set -x
mkdir -p gpurun_out/rc18
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/rc18/pb -o b -- python /root/repo/tools/time_phase_b.py graph > /root/repo/gpurun_out/rc18/pb.log 2>&1)
python - <<'PY'
import csv, glob
tr = list(csv.DictReader(open(glob.glob('/root/repo/gpurun_out/rc18/pb/**/*kernel_trace.csv', recursive=True)[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'scatter_accum' in r['Kernel_Name']]
seg = tr[idx[-2] + 1:idx[-1] + 1]
out = open('/root/repo/gpurun_out/rc18/phase_b_step_kernels.csv', 'w'); out.write('index,duration_us,kernel\n')
for i, r in enumerate(seg): out.write('%d,%.1f,"%s"\n' % (i, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:120].replace('"', "'")))
PY
rm -rf gpurun_out/rc18/pb
