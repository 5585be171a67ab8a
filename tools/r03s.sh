OUT=gpurun_out/r03s; mkdir -p $OUT
timeout 240 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_solver.py -m gpu -q -x --durations=5 > $OUT/pytest_subset.log 2>&1; tail -12 $OUT/pytest_subset.log
timeout 400 python tools/ab_bench.py --steps 2 new= new2= > $OUT/ab_windows.jsonl 2> $OUT/ab.err; cut -c1-330 $OUT/ab_windows.jsonl
