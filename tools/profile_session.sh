#!/bin/bash
# One profiling session on the GPU box: un-profiled bench, rocprofv3 kernel stats, two PMC passes (FETCH_SIZE / WRITE_SIZE separately,
# kernel-trace only).  Outputs under gpurun_out/sess/; tools/make_profile_summary.py turns them into profiles/<name>_summary.md.
set -u
R=/root/repo; O=$R/gpurun_out/sess; mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/pw.log 2>&1
tail -1 $O/bench.json | cut -c1-200
ls $O/stats/* $O/pf/* $O/pw/* | head -20
