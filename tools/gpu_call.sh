#!/bin/bash
# One gpurun call, by name (replaces the one-shot tools/r0*.sh scripts of earlier rounds; results land in gpurun_out/<tag>/ and the
# summaries worth keeping are copied to profiles/ by hand).
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <what> <tag> [args...]'
WHAT=$1; TAG=${2:-r05}; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
case $WHAT in
  pytest)       # pytest <tag> <pytest args...>
    timeout 1500 python -m pytest -m gpu -q "$@" > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log ;;
  ab)           # ab <tag> <steps> NAME=ENV=v,... ...      (tools/ab_bench.py)
    STEPS=$1; shift
    timeout 1200 python tools/ab_bench.py --steps $STEPS "$@" > $OUT/ab.jsonl 2> $OUT/ab.err; cut -c1-600 $OUT/ab.jsonl; tail -3 $OUT/ab.err ;;
  bench)        # bench <tag> [bench.py args]: the driver's command + a rocprofv3 kernel trace of the same command (secondary
                # configurations included, so that every figure of the line has its kernels in the stats; CPU baseline left out)
    timeout 900 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; tail -c 800 $OUT/bench.json; tail -5 $OUT/bench.err
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
    find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
    rm -rf $OUT/prof; head -12 $OUT/bench_kernel_stats.csv ;;
  trace)        # trace <tag> <script> [args]: rocprofv3 kernel trace + stats of a python script
    S=$1; shift
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/$S "$@" > $GRAFT_REPO_ROOT/$OUT/trace_stdout.txt 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
    find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
    rm -rf $OUT/prof; head -15 $OUT/kernel_stats.csv; tail -5 $OUT/trace_stdout.txt ;;
  pmc)          # pmc <tag> <script> [args]: FETCH_SIZE and WRITE_SIZE passes (separate runs) of a python script
    S=$1; shift
    for C in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o p -- python $GRAFT_REPO_ROOT/$S "$@" > $GRAFT_REPO_ROOT/$OUT/pmc_$C.log 2>&1)
      find $OUT/pmc_$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_${C}_counter_collection.csv
      rm -rf $OUT/pmc_$C
    done ;;
  sh)           # sh <tag> <command line>: anything else, output kept
    bash -c "$*" > $OUT/sh.log 2>&1; tail -30 $OUT/sh.log ;;
esac
