#!/bin/bash
# SQ counters of the default launch (k_solve_tw, assembly and selection fused): bash tools/pmc_tw.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/${tag}_sq_tw_$i -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-lod-system > $O/${tag}_sq_tw_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_sum.py /tmp/${tag}_sq_tw_1 /tmp/${tag}_sq_tw_2 /tmp/${tag}_sq_tw_3 > $O/${tag}_sq_tw.txt
grep -A1 "k_solve_tw" $O/${tag}_sq_tw.txt
