OUT=gpurun_out/r02s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_fullsize.py tests/test_gpu_solver.py tests/test_gpu_gen.py tests/test_gpu_sharded.py -m gpu -q > $OUT/pytest_subset.log 2>&1; tail -6 $OUT/pytest_subset.log
timeout 900 python tools/ab_bench.py --steps 3 win2= win1=MISPEC_DIA_WIN2=0 win2b= win1b=MISPEC_DIA_WIN2=0 > $OUT/ab.jsonl 2>&1; cut -c1-330 $OUT/ab.jsonl
