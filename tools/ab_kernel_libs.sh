#!/bin/bash
# per-kernel average durations of one C2 step under two builds of the library:  LIBS="a.so b.so" KERN="upconv|fir44" bash tools/ab_kernel_libs.sh
cd /root/repo
for lib in ${LIBS:-libeg3d_hip.so}; do
  (cd /tmp && export TMPDIR=/tmp && EG3D_LIBNAME=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/abl -o k -- python /root/repo/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-side --no-final-psnr --no-roofline > /dev/null 2>&1)
  echo "== $lib"
  python - <<PY
import csv,glob,re
f=glob.glob('/root/repo/gpurun_out/abl/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if re.search(r'${KERN:-upconv|fir44}', r['Name']): print('   ', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us  total', round(float(r['TotalDurationNs'])/1e6,2),'ms')
PY
  rm -rf gpurun_out/abl
done
