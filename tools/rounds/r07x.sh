#!/bin/bash
# round 4, final GPU call: the whole GPU suite, the driver's bench command (20 steps), the same command under rocprofv3
OUT=gpurun_out/r07x; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
T0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench wall seconds: $(( $(date +%s) - T0 ))"; tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof; head -12 $OUT/bench_kernel_stats.csv | cut -c1-220
