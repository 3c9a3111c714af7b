#!/bin/bash
# SQ / SQC counters of k_solve_nd (selection stage as its own launch): bash tools/pmc_nd.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES SQC_ICACHE_BUSY_CYCLES" ; do
  i=$((i+1))
  SLOD_SOLVE=nd SLOD_FUSE_SELECT=0 timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_sq_nd_$i -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $O/${tag}_sq_nd_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_sum.py $O/${tag}_sq_nd_1 $O/${tag}_sq_nd_2 $O/${tag}_sq_nd_3
