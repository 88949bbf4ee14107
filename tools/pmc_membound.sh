#!/bin/bash
# Counter passes over ONE eager C2 step sequence (bench.py --no-graph, 3 steps) for the memory-bound passes (upconv_epilogue, epilogue_bwd, fir44_adjoint_split,
# split_act_lds, upfirdn2d, torgb_mid*): where does the time go -- waiting (SQ_WAIT_ANY), issue stalls, the texture path, L2 hits, write stalls at the EA.
# One small group per pass (kernel-trace only).  CSV under gpurun_out/pmc_mem/;  python tools/pmc_summary.py gpurun_out/pmc_mem upconv_epilogue ... prints the means.
R=/root/repo; O=$R/gpurun_out/pmc_mem; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-final-psnr --no-side-configs --no-roofline --no-graph --repeats 1"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "MemUnitStalled L2CacheHit MemUnitBusy WriteUnitStalled"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python $R/bench.py $Q > $O/g$i.log 2>&1
  tail -n 2 $O/g$i.log | cut -c1-200
done
cd $R && python tools/pmc_summary.py $O upconv_epilogue epilogue_bwd fir44_adjoint_split split_act_lds upfirdn2d_nhwc4 torgb_mid scatter_accum16p gather_rows decode_rows > $O/summary.txt 2>&1
wc -l $O/summary.txt
