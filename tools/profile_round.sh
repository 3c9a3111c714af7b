#!/bin/bash
# Round-end measurement: bench line, rocprofv3 kernel stats, HBM counters (separate PMC passes).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 50 --warmup 5 > $R/gpurun_out/${tag}_bench.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-pipeline > $R/gpurun_out/${tag}_prof.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline > $R/gpurun_out/${tag}_pmc_$c.log 2>&1 || exit 1
done
tail -1 $R/gpurun_out/${tag}_bench.log
