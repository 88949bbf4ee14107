set -x
mkdir -p gpurun_out/rc10
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/rc10/vgg -o k -- python /root/repo/bench.py --loss-net vgg16 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-side-configs --no-final-psnr --no-roofline > /root/repo/gpurun_out/rc10/vgg.log 2>&1)
python - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob('/root/repo/gpurun_out/rc10/vgg/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_apply_norm' in r['Kernel_Name']]
seg = rows[idx[-2] + 1:idx[-1] + 1]
out = open('/root/repo/gpurun_out/rc10/vgg_step_kernels.csv', 'w')
out.write('index,duration_us,grid,kernel\n')
for i, r in enumerate(seg):
    out.write('%d,%.1f,%s,"%s"\n' % (i, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size', ''), r['Kernel_Name'][:110].replace('"', "'")))
print(len(seg), sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3)
PY
rm -rf gpurun_out/rc10/vgg
