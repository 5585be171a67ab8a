#!/bin/bash
# round 6, GPU call 16: the Cholesky halves of the wave-per-chunk solve (wide-band SparseCholesky)
OUT=gpurun_out/r11p; mkdir -p $OUT
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_geigs.py -k "banded_cholesky" > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
MISPEC_SHIFT=wave=0 timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_geigs.py -k "banded_cholesky_beyond" > $OUT/pytest_lane.log 2>&1; tail -5 $OUT/pytest_lane.log
