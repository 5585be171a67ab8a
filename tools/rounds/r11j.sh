#!/bin/bash
# round 6, GPU call 10: csr.hip split into three translation units + the shift solve's host-side set_shift work on threads:
# SpMV / windows / shift / solver modules, set_shift phases of C5 and W5 again, shard-size trace with its gaps (final host turn)
OUT=gpurun_out/r11j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest -m gpu -q -x tests/test_gpu_spmv.py tests/test_gpu_windows.py tests/test_gpu_shift.py tests/test_gpu_geigs.py tests/test_gpu_fullsize.py tests/test_gpu_solver.py tests/test_gpu_sharded.py tests/test_gpu_reorder.py tests/test_gpu_staged.py tests/test_gpu_tiles.py tests/test_gpu_gen.py > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
MISPEC_SHIFT=profile=1 python tools/bench_configs.py c5 > $OUT/c5.json 2> $OUT/c5_set_shift_phases.txt; tail -9 $OUT/c5_set_shift_phases.txt; cut -c1-600 $OUT/c5.json
MISPEC_SHIFT=profile=1 python tools/bench_configs.py w5 > $OUT/w5.json 2> $OUT/w5_set_shift_phases.txt; tail -12 $OUT/w5_set_shift_phases.txt; cut -c1-500 $OUT/w5.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/c2_solves.py --size 1250000 --solves 5 > $GRAFT_REPO_ROOT/$OUT/shard_only_stdout.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_1250000_rows.csv
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T k_orth_lagged > $OUT/trace_gaps_1250000_rows.txt; head -12 $OUT/trace_gaps_1250000_rows.txt | cut -c1-150
rm -rf $OUT/prof; head -12 $OUT/kernel_stats_1250000_rows.csv | cut -c1-200; tail -1 $OUT/shard_only_stdout.json | cut -c1-400
