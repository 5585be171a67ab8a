#!/bin/bash
# round 6, GPU call 17: any StorageIndex at the sparse operators (C++ drop-in program), the reference's own test programs, small-solve latency
OUT=gpurun_out/r11q; mkdir -p $OUT
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_cpp_dropin.py tests/test_gpu_reference_programs.py > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python tools/small_latency.py > $OUT/small_solve_latency.jsonl 2> $OUT/small.err; cut -c1-260 $OUT/small_solve_latency.jsonl
