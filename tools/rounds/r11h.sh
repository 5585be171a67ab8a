#!/bin/bash
# round 6, GPU call 8: the driver's bench command on the current tree (+ its rocprofv3 kernel trace), the multi-process module
# with the new 8-rank bench test, the small-kernel module with the pipelined kernel behind small=device
OUT=gpurun_out/r11h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_multiproc.py tests/test_gpu_small.py > $OUT/pytest_multiproc_small.log 2>&1; tail -5 $OUT/pytest_multiproc_small.log
MISPEC_SMALL=device timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_solver.py tests/test_gpu_onesweep.py -k "not fullsize" > $OUT/pytest_small_device.log 2>&1; tail -3 $OUT/pytest_small_device.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -5 $OUT/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof; head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
