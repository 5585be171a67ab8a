#!/bin/bash
# round 4, GPU call: device-driven / one-sweep steps for the Cholesky mode of the generalized problem
OUT=gpurun_out/r07w; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_geigs.py tests/test_gpu_cpp_dropin.py tests/test_gpu_reference_programs.py -k "not Davidson" > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
