OUT=gpurun_out/r02g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_geigs.py tests/test_gpu_shift.py tests/test_gpu_sharded.py tests/test_gpu_tiles.py tests/test_gpu_spmv.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -8 $OUT/pytest_subset.log
