#!/bin/bash
# usage (on the GPU box): tools/pmc_conv.sh <tag> <bench_conv args...> : SQ counter passes over one conv microbench
cd /tmp && export TMPDIR=/tmp
R=/root/repo
tag=$1; shift
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_INSTS_BRANCH"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_${tag}_$n -o p --output-format csv -- python3 $R/tools/bench_conv.py "$@" --reps 3 > $R/gpurun_out/pmc_${tag}_$n.log 2>&1 || exit 1
done
