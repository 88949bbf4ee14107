#!/bin/bash
# kernels of ONE replayed C2 step (index, duration, name) under a list of environment settings, from kernel traces taken in the same GPU call:
#   ENVS="EG3D_CONV_WS_S2=0;EG3D_CONV_WS_S2=1" PAT="conv_ws|conv_igemm_kernel<32|Li32ELi128" bash tools/ab_ws_s2.sh
cd /root/repo
IFS=';' read -ra LIST <<< "${ENVS:-X=1}"
k=0
for e in "${LIST[@]}"; do
  k=$((k+1))
  (cd /tmp && export TMPDIR=/tmp && env $e timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/abs$k -o k -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side --no-final-psnr --no-roofline > /dev/null 2>&1)
  E="$e" K=$k PAT="${PAT:-conv_ws|conv_igemm_kernel<32|Li32ELi128}" python - <<'PY'
import csv, glob, os, re
k = os.environ['K']
rows = list(csv.DictReader(open(glob.glob(f'/root/repo/gpurun_out/abs{k}/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_apply_norm' in r['Kernel_Name']]
seg = rows[idx[-2] + 1:idx[-1] + 1]
print(os.environ['E'], ': kernels', len(seg), 'busy us', sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3)
for i, r in enumerate(seg):
    if re.search(os.environ['PAT'], r['Kernel_Name']):
        print('   ', i, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:90])
PY
  rm -rf gpurun_out/abs$k
done
