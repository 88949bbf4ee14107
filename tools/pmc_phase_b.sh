#!/bin/bash
# PMC passes over the eager Phase-B step (tools/time_phase_b.py): each counter group in its own run, --kernel-trace only.
# Outputs under gpurun_out/pmcb/g*/ ; tools/pmc_summary.py gpurun_out/pmcb <kernel substrings> prints per-kernel means.
set -u
R=/root/repo; O=$R/gpurun_out/pmcb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/g1 -o f -- python $R/tools/time_phase_b.py eager > $O/g1.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/g2 -o w -- python $R/tools/time_phase_b.py eager > $O/g2.log 2>&1
timeout 300 rocprofv3 --pmc MfmaUtil LdsUtil --kernel-trace --output-format csv -d $O/g3 -o m -- python $R/tools/time_phase_b.py eager > $O/g3.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmcb conv_wgrad rows_gram decode_rows adam_step weight_grad_finish pack_conv > gpurun_out/pmcb_summary.txt 2>&1
rm -f $O/g*/*kernel_trace.csv
tail -60 gpurun_out/pmcb_summary.txt
