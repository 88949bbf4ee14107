#!/bin/bash
# Counter passes over the forward renderer (tools/time_renderer.py): instruction mix, wait cycles, texture-address unit.
# One small group per pass (kernel-trace only), CSV under gpurun_out/pmc_render/.
R=/root/repo; O=$R/gpurun_out/pmc_render; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" \
           "TA_BUSY TA_TOTAL_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python $R/tools/time_renderer.py > $O/g$i.log 2>&1
done
ls $O
