#!/bin/bash
# SQ counter passes (separate rocprofv3 runs, kernel-trace only).  usage: pmc_sq.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MUL_F64" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sq${tag}_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/sq${tag}_$i.log 2>&1 || exit 1
done
