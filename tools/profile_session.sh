#!/bin/bash
# One profiling session on the GPU box: un-profiled bench, rocprofv3 kernel stats, PMC passes (FETCH_SIZE / WRITE_SIZE / SQ utilisation
# counters, each in its own run with --kernel-trace only).  Outputs under gpurun_out/sess/; tools/make_profile_summary.py turns them
# into profiles/<name>_summary.md + profiles/traffic_table.json.   usage: tools/profile_session.sh [extra bench flags]
set -u
R=/root/repo; O=$R/gpurun_out/sess; rm -rf $O; mkdir -p $O
X="$*"
cd $R && python bench.py $X > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-final-psnr --no-side-configs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 10 --warmup 2 $Q $X > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o f -- python $R/bench.py --steps 3 --warmup 1 $Q --no-roofline --no-graph $X > $O/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o w -- python $R/bench.py --steps 3 --warmup 1 $Q --no-roofline --no-graph $X > $O/pw.log 2>&1
timeout 300 rocprofv3 --pmc MfmaUtil LdsUtil --kernel-trace --output-format csv -d $O/pm -o m -- python $R/bench.py --steps 3 --warmup 1 $Q --no-roofline --no-graph $X > $O/pm.log 2>&1
timeout 300 rocprofv3 --pmc LdsBankConflict SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $O/pl -o l -- python $R/bench.py --steps 3 --warmup 1 $Q --no-roofline --no-graph $X > $O/pl.log 2>&1
tail -1 $O/bench.json | cut -c1-300
ls $O/stats/* $O/pf/* $O/pw/* $O/pm/* $O/pl/* | head -30
for f in pm pl; do tail -n 3 $O/$f.log; done
