OUT=gpurun_out/r02c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py tests/test_gpu_sharded.py tests/test_gpu_multiproc.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -5 $OUT/pytest_subset.log
MISPEC_TILES_SYNC=2 timeout 600 python -m pytest tests/test_gpu_tiles.py -m gpu -q > $OUT/pytest_tiles_sync2.log 2>&1; tail -3 $OUT/pytest_tiles_sync2.log
for s in 0 1 2 4 8; do MISPEC_TILES_SYNC=$s timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err; done; cat $OUT/mrand.jsonl
bash tools/pmc_pass.sh $OUT l2_sync0 "TCC_HIT_sum TCC_MISS_sum" tools/pmc_probe_mrand.py
MISPEC_TILES_SYNC=2 bash tools/pmc_pass.sh $OUT l2_sync2 "TCC_HIT_sum TCC_MISS_sum" tools/pmc_probe_mrand.py
bash tools/pmc_pass.sh $OUT fetch_sync0 "FETCH_SIZE" tools/pmc_probe_mrand.py
