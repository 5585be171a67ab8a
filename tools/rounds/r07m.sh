#!/bin/bash
# round 4, GPU call: device-driven / one-sweep steps for the shift-solve, dense and device-pointer operators
OUT=gpurun_out/r07m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_shift.py tests/test_gpu_dense.py tests/test_gpu_reference_programs.py tests/test_gpu_cpp_dropin.py tests/test_gpu_fac.py tests/test_gpu_gen.py tests/test_gpu_geigs.py > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
for M in onesweep reference; do
  MISPEC_ORTH=$M timeout 300 python tools/bench_configs.py c5 > $OUT/c5_$M.json 2> $OUT/c5_$M.err; cut -c1-700 $OUT/c5_$M.json; tail -2 $OUT/c5_$M.err
done
MISPEC_HOST_STEPS=1 timeout 300 python tools/bench_configs.py c5 > $OUT/c5_hoststeps.json 2>&1; cut -c1-400 $OUT/c5_hoststeps.json
