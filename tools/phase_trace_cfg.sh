#!/bin/bash
# like phase_trace.sh for another configuration. usage: phase_trace_cfg.sh <tag> <config> <mask>...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; cfg=$2; shift; shift
export SLOD_LIB_PATH=$R/dealii-slod_amd/lib/libslod_hip_diag.so SLOD_FUSE_SELECT=0 SLOD_FUSE_ASSEMBLE=0
for mask in "$@"; do
  export SLOD_DIAG=$mask
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pt${tag}_$mask -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/pt${tag}_$mask.log 2>&1 || exit 1
  grep -h "k_solve\|k_select" $R/gpurun_out/pt${tag}_$mask/*/*kernel_stats.csv | cut -d, -f1-4 | sed "s/^/mask $mask: /"
done
