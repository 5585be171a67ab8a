OUT=gpurun_out/r03q; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_solver.py -m gpu -q -x > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
timeout 900 python tools/ab_bench.py --steps 2 early= late=MISPEC_DIA_LATE_EPILOGUE=1 early2= late2=MISPEC_DIA_LATE_EPILOGUE=1 > $OUT/ab_epilogue.jsonl 2> $OUT/ab.err; cut -c1-330 $OUT/ab_epilogue.jsonl
