OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_sharded.py -m gpu -q -x > $OUT/pytest_subset.log 2>&1; tail -2 $OUT/pytest_subset.log
timeout 400 python tools/ab_bench.py --steps 2 early=MISPEC_SPMV_DIA=0,MISPEC_SPMV_CODES=0 late=MISPEC_SPMV_DIA=0,MISPEC_SPMV_CODES=0,MISPEC_SPMV_LATE_EPILOGUE=1 early_codes=MISPEC_SPMV_DIA=0 late_codes=MISPEC_SPMV_DIA=0,MISPEC_SPMV_LATE_EPILOGUE=1 > $OUT/ab_csr_early_epilogue.jsonl 2> $OUT/ab.err; cut -c1-200 $OUT/ab_csr_early_epilogue.jsonl
