OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py tests/test_gpu_geigs.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log
for bt in 32 16; do MISPEC_SHIFT_BATCH=$bt timeout 280 python tools/c5_probe.py >> $OUT/c5.jsonl 2>> $OUT/err.log; done
for bt in 16 8; do MISPEC_SHIFT_PREFETCH=1 MISPEC_SHIFT_BATCH=$bt timeout 280 python tools/c5_probe.py >> $OUT/c5.jsonl 2>> $OUT/err.log; done
MISPEC_SHIFT_LANES=32 timeout 280 python tools/c5_probe.py >> $OUT/c5.jsonl 2>> $OUT/err.log
MISPEC_SHIFT_CHUNK=160,128 timeout 280 python tools/c5_probe.py >> $OUT/c5.jsonl 2>> $OUT/err.log
MISPEC_SHIFT_CHUNK=96,128 timeout 280 python tools/c5_probe.py >> $OUT/c5.jsonl 2>> $OUT/err.log
cat $OUT/c5.jsonl
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o c5 -- python $GRAFT_REPO_ROOT/tools/c5_probe.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); cp $f $OUT/c5_kernel_stats.csv; rm -rf $OUT/prof; grep -E "chunk_solve|back_subst|block_gemv|sep_rhs|row_gemv" $OUT/c5_kernel_stats.csv | sed 's/(long.*)",/",/; s/(anonymous namespace):://g' | cut -c1-200
