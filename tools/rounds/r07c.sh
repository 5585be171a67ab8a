#!/bin/bash
# round 4, GPU call 3: the staged SpMV format — parity, then M-rand at full size against tiles and CSR, with a kernel trace
OUT=gpurun_out/r07c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_tiles.py tests/test_gpu_onesweep.py -q -x > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
MISPEC_SPMV_STAGED=1 timeout 600 python tools/bench_staged.py > $OUT/bench_staged.jsonl 2> $OUT/bench_staged.err
cat $OUT/bench_staged.jsonl; tail -3 $OUT/bench_staged.err
(cd /tmp && export TMPDIR=/tmp && MISPEC_SPMV_STAGED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_staged.py > $GRAFT_REPO_ROOT/$OUT/trace_stdout.txt 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/prof; head -12 $OUT/kernel_stats.csv | cut -c1-260
