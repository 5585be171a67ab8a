OUT=gpurun_out/r03p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_shift.py tests/test_gpu_geigs.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log; grep -n "^E " $OUT/pytest_subset.log | head -8 | cut -c1-500
timeout 280 python tools/c5_probe.py
