OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py tests/test_gpu_geigs.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log
for bi in 256 0; do MISPEC_SHIFT_BLOCK_INVERSE=$bi timeout 280 python tools/c5_probe.py >> $OUT/c5_block_inverse.jsonl 2>> $OUT/err.log; done
for ch in 64,128 96,128 64,128,0,4096 ; do MISPEC_SHIFT_CHUNK=$ch timeout 280 python tools/c5_probe.py >> $OUT/c5_block_inverse.jsonl 2>> $OUT/err.log; done
cat $OUT/c5_block_inverse.jsonl
cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/c5_probe.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err; cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1); cp $f $OUT/c5_kernel_stats.csv; head -16 $OUT/c5_kernel_stats.csv | cut -c1-220
