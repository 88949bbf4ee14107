#!/bin/bash
# per-kernel times of the renderer loop under a list of environment settings:  ENVS="A=1 B=2;A=2" KERN=scatter bash tools/ab_libs.sh
cd /root/repo
IFS=';' read -ra LIST <<< "${ENVS:-;}"
for e in "${LIST[@]}"; do
  echo "== $e"
  (cd /tmp && export TMPDIR=/tmp && env $e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/scx -o s -- python /root/repo/tools/time_renderer.py > /root/repo/gpurun_out/scx.log 2>&1)
  grep "fwd+bwd" gpurun_out/scx.log | head -1
  python - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/scx/**/*_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if '${KERN:-scatter}' in r['Name']: print('   ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
  rm -rf gpurun_out/scx
done
