#!/bin/bash
# round 6, GPU call 14: wave-per-chunk solve, third version (shifted factor copy, straight-line batches, masks at use): tests, timings, trace
OUT=gpurun_out/r11n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_shift.py tests/test_gpu_geigs.py > $OUT/pytest_shift_geigs.log 2>&1; tail -6 $OUT/pytest_shift_geigs.log
for B in 32 12 64; do W5_B=$B python tools/bench_configs.py w5 2>$OUT/w5_b$B.err | tee -a $OUT/w5_wave.jsonl | cut -c1-330; done
(cd /tmp && W5_B=32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_configs.py w5 > $GRAFT_REPO_ROOT/$OUT/w5_traced.json 2> $GRAFT_REPO_ROOT/$OUT/w5_traced.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/w5_b32_kernel_stats.csv
rm -rf $OUT/prof; head -6 $OUT/w5_b32_kernel_stats.csv | cut -c1-200
python tools/bench_configs.py c5 2>/dev/null | cut -c1-420
