OUT=gpurun_out/r03h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py tests/test_gpu_geigs.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log
for bt in 32 16 8; do MISPEC_SHIFT_BATCH=$bt timeout 280 python tools/c5_probe.py >> $OUT/c5_sched.jsonl 2>> $OUT/err.log; done
MISPEC_SHIFT_LDS=0 timeout 280 python tools/c5_probe.py >> $OUT/c5_sched.jsonl 2>> $OUT/err.log
MISPEC_SHIFT_BATCH=16 MISPEC_SHIFT_LANES=32 timeout 280 python tools/c5_probe.py >> $OUT/c5_sched.jsonl 2>> $OUT/err.log
MISPEC_SHIFT_BATCH=16 MISPEC_SHIFT_CHUNK=64,128 timeout 280 python tools/c5_probe.py >> $OUT/c5_sched.jsonl 2>> $OUT/err.log
cat $OUT/c5_sched.jsonl
(cd /tmp && export TMPDIR=/tmp && MISPEC_SHIFT_BATCH=16 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o c5 -- python $GRAFT_REPO_ROOT/tools/c5_probe.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); cp $f $OUT/c5_kernel_stats.csv; rm -rf $OUT/prof; head -8 $OUT/c5_kernel_stats.csv | cut -c1-200
