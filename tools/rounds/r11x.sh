#!/bin/bash
# round 6, GPU call 24: the host-side factorisation with 1 / 3 / all host threads (same bits)
OUT=gpurun_out/r11x; mkdir -p $OUT
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_shift.py -k "thread_count or wave_per_chunk" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
