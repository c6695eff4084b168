#!/bin/bash
# usage (GPU box): tools/pmc_sets.sh <variant|default> <outfile> [kbench args...]
# Four counter passes (one block of counters each, every pass under timeout) over the iteration kernel of one library build.
v=$1; out=$2; shift; shift
if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so; fi
: > $GRAFT_REPO_ROOT/$out
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_BUSY_CYCLES" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  $GRAFT_REPO_ROOT/tools/pmc_any.sh pmcset_${v}_$i "$set" iterate_kernel $GRAFT_REPO_ROOT/tools/kbench.py "$@" > /dev/null 2>&1
  cat $GRAFT_REPO_ROOT/gpurun_out/pmcset_${v}_${i}_pmc.txt >> $GRAFT_REPO_ROOT/$out
done
cat $GRAFT_REPO_ROOT/$out
