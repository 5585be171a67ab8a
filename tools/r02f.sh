OUT=gpurun_out/r02f; mkdir -p $OUT
for x in 0 1 2; do MISPEC_TILES_XLOAD=$x timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err; done
cat $OUT/mrand.jsonl
timeout 300 python -m pytest tests/test_gpu_shift.py -m gpu -q > $OUT/pytest_shift.log 2>&1; tail -3 $OUT/pytest_shift.log
