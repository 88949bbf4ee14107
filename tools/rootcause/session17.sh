set -x
mkdir -p gpurun_out/rc17
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/rc17/clk -o c -- python /root/repo/tools/rootcause/clock_probe.py > /root/repo/gpurun_out/rc17/clk.log 2>&1)
python - <<'PY'
import csv, glob, collections
d = '/root/repo/gpurun_out/rc17/clk'
tr = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]))}
agg = collections.defaultdict(list); seen = collections.Counter()
for r in csv.DictReader(open(glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0])):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    t = tr.get(r['Dispatch_Id'])
    if not t: continue
    ns = int(t['End_Timestamp']) - int(t['Start_Timestamp'])
    nm = r['Kernel_Name'][:60]; seen[nm] += 1
    if 'mfma_probe' in nm: nm += ' [random data]' if seen[nm] <= 20 else ' [zeros]'
    agg[nm].append((float(r['Counter_Value']), ns))
for k, v in agg.items():
    if len(v) < 5: continue
    v = v[len(v) // 2:]                      # second half of the launches: the clock has settled
    cyc = sum(a for a, b in v) / len(v); ns = sum(b for a, b in v) / len(v)
    print('%-72s launches %3d  cycles %12.0f  us %9.1f  -> %.3f GHz (counter / 8 XCDs: %.3f)' % (k, len(v), cyc, ns / 1e3, cyc / ns, cyc / ns / 8))
PY
rm -rf gpurun_out/rc17/clk
