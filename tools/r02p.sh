OUT=gpurun_out/r02p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_reorder.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
MISPEC_SPMV_SMALL_CHUNK=1 timeout 600 python -m pytest tests/test_gpu_spmv.py -m gpu -q > $OUT/pytest_small.log 2>&1; tail -2 $OUT/pytest_small.log
for k in 0 1; do MISPEC_SPMV_SMALL_CHUNK=$k timeout 600 python tools/bench_stencil.py 215 RCM >> $OUT/stencil_rcm.jsonl 2>> $OUT/err.log; done; cut -c1-420 $OUT/stencil_rcm.jsonl
