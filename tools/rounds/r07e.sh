#!/bin/bash
# round 4, GPU call 5: staged format, phase 2 with one chunk per wavefront and two-stage prefetch
OUT=gpurun_out/r07e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_staged.py -q -x > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
(cd /tmp && export TMPDIR=/tmp && BENCH_FORMATS=4,3 MISPEC_SPMV_STAGED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_staged.py > $GRAFT_REPO_ROOT/$OUT/trace_stdout.txt 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/prof; grep -E "staged|tiles" $OUT/kernel_stats.csv | cut -c1-60,200-330; cut -c1-400 $OUT/trace_stdout.txt
