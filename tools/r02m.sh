OUT=gpurun_out/r02m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fac.py tests/test_gpu_spmv.py tests/test_gpu_svd.py tests/test_gpu_geigs.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -5 $OUT/pytest_subset.log
timeout 900 python tools/ab_bench.py --steps 3 fused= unfused=MISPEC_FUSE_SCALE=0 fused2= unfused2=MISPEC_FUSE_SCALE=0 > $OUT/ab.jsonl 2>&1; cat $OUT/ab.jsonl
