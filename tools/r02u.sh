OUT=gpurun_out/r02u; mkdir -p $OUT
timeout 300 python tools/bench_mrand.py 1e7 > $OUT/mrand.jsonl 2> $OUT/err.log
MISPEC_TILES_SYNC=0 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
MISPEC_TILES_SYNC=10 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
cat $OUT/mrand.jsonl
timeout 600 python -m pytest tests/test_gpu_tiles.py -m gpu -q > $OUT/pytest_tiles.log 2>&1; tail -2 $OUT/pytest_tiles.log
