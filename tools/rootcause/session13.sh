set -x
mkdir -p gpurun_out/rc13
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/rc13/pn -o k -- python /root/repo/tools/time_pose_net.py > /root/repo/gpurun_out/rc13/pose.log 2>&1)
tail -3 gpurun_out/rc13/pose.log
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/rc13/pn/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print('%6d %9.1f us avg %7.2f ms total %5.1f%%  %s' % (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot, r['Name'][:100]))
PY
rm -rf gpurun_out/rc13/pn
