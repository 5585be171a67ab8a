#!/bin/bash
# One gpurun call of a round: GPU tests, the bench line, a rocprofv3 kernel trace of the same command, the PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a [tests|notests] [pmc|nopmc]'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
  tail -15 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; tail -5 $OUT/bench.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --no-secondary --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
head -8 $OUT/bench_kernel_stats.csv
if [ "${3:-pmc}" = "pmc" ]; then
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -o fetch -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc -o write -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1)
  F=$(find $OUT/pmc -name "fetch_counter_collection.csv" | head -1); W=$(find $OUT/pmc -name "write_counter_collection.csv" | head -1)
  python tools/pmc_summarize.py $F $W 10000000 $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
  cp $F $OUT/pmc_fetch_counter_collection.csv; cp $W $OUT/pmc_write_counter_collection.csv
  rm -rf $OUT/pmc
  grep -i "spmv\|calibration" $OUT/pmc_summary.txt | head
fi
rm -rf $OUT/prof
timeout 300 python tools/two_ranks_one_gpu.py $OUT/two_ranks_one_gpu_rccl.json > /dev/null 2>&1
head -c 400 $OUT/two_ranks_one_gpu_rccl.json
