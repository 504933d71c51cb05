#!/bin/bash
# PMC passes over the stepping kernels (separate runs per counter group, kernel-trace only).  Usage: gpu_pmc_passes.sh OUTPREFIX
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_WRITE_REQ_sum" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 $R/scripts/gpu_pmc_step.py > $O/$1_pmc_$i.log 2>&1
  echo "== $grp" >> $O/$1_pmc.txt
  python3 $R/scripts/summarize_pmc.py /tmp/pmc_$i k_narrow k_pipe k_step >> $O/$1_pmc.txt 2>&1
done
cat $O/$1_pmc.txt
