OUT=gpurun_out/r03b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py tests/test_gpu_geigs.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log
for pp in 1 0 1 0; do MISPEC_SHIFT_PIPE=$pp timeout 280 python tools/c5_probe.py >> $OUT/c5_pipe.jsonl 2>> $OUT/err.log; done; cat $OUT/c5_pipe.jsonl
