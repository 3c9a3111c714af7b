#!/bin/bash
# Round-3 measurement set (on the GPU box, from the repo root): bash tools/profile_round3.sh <tag>
# bench lines of C2/C4/C3 (default kernels), rocprofv3 kernel stats of C2, HBM counters of C2.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
timeout -k 10 300 $B --steps 200 --warmup 20 > $O/${tag}_bench_C2.log 2>&1 || exit 1
timeout -k 10 300 $B > $O/${tag}_bench_C2_default.log 2>&1 || exit 1
timeout -k 10 300 $B --config C4 --steps 20 --warmup 3 --no-lod-system > $O/${tag}_bench_C4.log 2>&1 || exit 1
timeout -k 10 600 $B --config C3 --steps 3 --warmup 1 --no-lod-system > $O/${tag}_bench_C3.log 2>&1 || exit 1
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_prof_C2 -- $B --steps 50 --warmup 5 --no-cpu-baseline --no-pipeline --no-lod-system > $O/${tag}_prof_C2.log 2>&1 || exit 1
cp $(find /tmp/${tag}_prof_C2 -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats_C2.csv
echo "kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/${tag}_pmc_$c -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-lod-system > $O/${tag}_pmc_$c.log 2>&1 || exit 1
  python3 $R/tools/pmc_sum.py /tmp/${tag}_pmc_$c > $O/${tag}_pmc_$c.txt
done
echo "hbm counters done"
tail -1 $O/${tag}_bench_C2.log | cut -c1-400
