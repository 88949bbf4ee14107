#!/bin/bash
# A/B of the plane-gradient scatter variants (EG3D_SCATTER=1|2|3|4) inside one GPU call: parity tests + per-kernel times of the renderer loop.
cd /root/repo
for cfg in ${CFGS:-"3:0 4:0"}; do
  v=${cfg%%:*}; d=${cfg##*:}
  echo "== EG3D_SCATTER=$v DBG=$d"
  if [ "$d" = "0" ]; then EG3D_SCATTER=$v timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "scatter or render or triplane" -x 2>&1 | tail -1; fi
  (cd /tmp && export TMPDIR=/tmp && EG3D_SCATTER_DBG=$d EG3D_SCATTER=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/sc$v$d -o s -- python /root/repo/tools/time_renderer.py > /root/repo/gpurun_out/sc$v$d.log 2>&1)
  grep "ms" gpurun_out/sc$v$d.log | head -1
  python - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/sc$v$d/**/*_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scatter' in r['Name']: print('   ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
  rm -rf gpurun_out/sc$v$d
done
