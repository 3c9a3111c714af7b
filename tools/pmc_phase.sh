#!/bin/bash
# VALU instruction count per phase of the patch solve: the diag library with stages unfused and
# SLOD_DIAG phase-skipping masks, one SQ counter pass each.  usage: pmc_phase.sh <tag> <mask>...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
export SLOD_LIB_PATH=$R/dealii-slod_amd/lib/libslod_hip_diag.so SLOD_FUSE_SELECT=0 SLOD_FUSE_ASSEMBLE=0
for mask in "$@"; do
  export SLOD_DIAG=$mask
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/ph${tag}_$mask -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/ph${tag}_$mask.log 2>&1 || exit 1
done
