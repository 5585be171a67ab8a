#!/bin/bash
# round 6, GPU call 12: the wave-per-chunk solve of wide bands: shift / geigs modules, W5 timings (b = 32, 12, 64)
OUT=gpurun_out/r11l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_shift.py tests/test_gpu_geigs.py > $OUT/pytest_shift_geigs.log 2>&1; tail -12 $OUT/pytest_shift_geigs.log
for B in 32 12 64; do W5_B=$B python tools/bench_configs.py w5 2>$OUT/w5_b$B.err | tee -a $OUT/w5_wave.jsonl | cut -c1-420; done
W5_B=32 MISPEC_SHIFT=wave=0 python tools/bench_configs.py w5 2>>$OUT/w5_lane.err | tee -a $OUT/w5_lane.jsonl | cut -c1-420
