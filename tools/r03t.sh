OUT=gpurun_out/r03t; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_tiles.py -m gpu -q -x > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
for pp in 1 0; do MISPEC_TILES_PIPELINE=$pp timeout 200 python tools/bench_mrand.py 1e7 >> $OUT/mrand_pipeline.jsonl 2>> $OUT/err.log; done; cut -c1-400 $OUT/mrand_pipeline.jsonl
