OUT=gpurun_out/r02e; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_shift.py -m gpu -q -k "zero_leading" > $OUT/pytest_zero.log 2>&1; grep -E "^E|passed|failed" $OUT/pytest_zero.log | tail -5
for s in 1 2 4; do MISPEC_TILES_SYNC=$s timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err; done
MISPEC_TILES_SYNC=2 MISPEC_TILES_WG_PER_CU=5 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err
cat $OUT/mrand.jsonl; tail -3 $OUT/mrand.err
MISPEC_TILES_SYNC=2 timeout 600 python -m pytest tests/test_gpu_tiles.py -m gpu -q > $OUT/pytest_tiles_sync2.log 2>&1; tail -3 $OUT/pytest_tiles_sync2.log
