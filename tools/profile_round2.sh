#!/bin/bash
# Round-2 measurement set (on the GPU box, from the repo root): bash tools/profile_round2.sh <tag>
# bench lines of C2/C4/C3, rocprofv3 kernel stats, HBM counters, SQ counters of the default (tw)
# and the opt-in matrix-pipe (mf) patch solve.  PMC passes are separate runs, kernel-trace only.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
timeout -k 10 300 $B --steps 200 --warmup 20 > $O/${tag}_bench_C2.log 2>&1 || exit 1
timeout -k 10 300 $B --config C4 --steps 20 --warmup 3 > $O/${tag}_bench_C4.log 2>&1 || exit 1
timeout -k 10 600 $B --config C3 --steps 3 --warmup 1 > $O/${tag}_bench_C3.log 2>&1 || exit 1
SLOD_SOLVE=mf timeout -k 10 300 $B --steps 100 --warmup 10 --no-cpu-baseline > $O/${tag}_bench_C2_mf.log 2>&1 || exit 1
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_C2 -- $B --steps 50 --warmup 5 --no-cpu-baseline --no-pipeline > $O/${tag}_prof_C2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_C4 -- $B --config C4 --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline > $O/${tag}_prof_C4.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_C3 -- $B --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $O/${tag}_prof_C3.log 2>&1 || exit 1
echo "kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${tag}_pmc_$c -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > $O/${tag}_pmc_$c.log 2>&1 || exit 1
done
echo "hbm counters done"
for solver in tw mf; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MUL_F64" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" ; do
    i=$((i+1))
    SLOD_SOLVE=$solver SLOD_FUSE_SELECT=0 SLOD_FUSE_ASSEMBLE=0 timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_sq_${solver}_$i -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $O/${tag}_sq_${solver}_$i.log 2>&1 || exit 1
  done
done
echo "sq counters done"
tail -1 $O/${tag}_bench_C2.log
