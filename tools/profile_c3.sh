cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 600 python3 $R/bench.py --config C3 --steps 3 --warmup 1 > $O/r02d_bench_C3.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02d_prof_C3 -- python3 $R/bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $O/r02d_prof_C3.log 2>&1 || exit 1
