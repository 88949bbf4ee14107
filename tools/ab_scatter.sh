#!/bin/bash
# A/B of the plane-gradient scatter variants (EG3D_SCATTER=1|3|4) inside one GPU call: parity tests + per-kernel times of the renderer loop.
#   VARIANTS="3 4" bash tools/ab_scatter.sh
cd /root/repo
for v in ${VARIANTS:-1 3 4}; do
  echo "== EG3D_SCATTER=$v"
  EG3D_SCATTER=$v timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "render" -x 2>&1 | tail -1
  (cd /tmp && export TMPDIR=/tmp && EG3D_SCATTER=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/sc$v -o s -- python /root/repo/tools/time_renderer.py > /root/repo/gpurun_out/sc$v.log 2>&1)
  grep "ms" gpurun_out/sc$v.log | head -1
  python - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/sc$v/**/*_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'scatter' in r['Name']: print('   ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
  rm -rf gpurun_out/sc$v
done
