OUT=gpurun_out/r03o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py -m gpu -q -k "variants" > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o c5 -- python $GRAFT_REPO_ROOT/tools/c5_probe.py > $GRAFT_REPO_ROOT/$OUT/c5_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); cp $f $OUT/c5_kernel_stats.csv; rm -rf $OUT/prof; grep -E "chunk_solve|back_subst|block_gemv|sep_rhs|row_gemv" $OUT/c5_kernel_stats.csv | sed 's/(long.*)",/",/; s/(anonymous namespace):://g' | cut -c1-200
