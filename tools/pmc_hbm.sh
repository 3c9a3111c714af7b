#!/bin/bash
# HBM counters of one bench run (two PMC passes). usage: pmc_hbm.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/hbm${tag}_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/hbm${tag}_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_sum.py $R/gpurun_out/hbm${tag}_FETCH_SIZE $R/gpurun_out/hbm${tag}_WRITE_SIZE | grep -A1 "k_solve"
