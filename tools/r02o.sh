OUT=gpurun_out/r02o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_multiproc.py -m gpu -q > $OUT/pytest_multiproc.log 2>&1; tail -30 $OUT/pytest_multiproc.log | cut -c1-300
