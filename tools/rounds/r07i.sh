#!/bin/bash
# round 4, GPU call 9: kernel trace of the driver's bench command; PMC traffic of the staged kernels (separate passes)
OUT=gpurun_out/r07i; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof; head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe_staged.py > $GRAFT_REPO_ROOT/$OUT/pmc_$C.log 2>&1)
  find $OUT/pmc_$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_${C}_counter_collection.csv
  rm -rf $OUT/pmc_$C
done
python tools/pmc_summarize.py $OUT/pmc_FETCH_SIZE_counter_collection.csv $OUT/pmc_WRITE_SIZE_counter_collection.csv 10000000 $OUT/pmc_traffic_mrand.json | head -20
