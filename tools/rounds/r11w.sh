#!/bin/bash
# round 6, GPU call 23: the option settings that are not the default: small=device (pipelined kernel), small=host-serial, orth_kernel=reg
OUT=gpurun_out/r11w; mkdir -p $OUT
MISPEC_SMALL=device timeout 900 python -m pytest -m gpu -q tests/test_gpu_solver.py tests/test_gpu_onesweep.py tests/test_gpu_gen.py > $OUT/pytest_small_device.log 2>&1; tail -3 $OUT/pytest_small_device.log
MISPEC_SMALL=host-serial timeout 900 python -m pytest -m gpu -q tests/test_gpu_solver.py tests/test_gpu_onesweep.py > $OUT/pytest_small_host_serial.log 2>&1; tail -3 $OUT/pytest_small_host_serial.log
MISPEC_ORTH_KERNEL=reg timeout 1200 python -m pytest -m gpu -q tests/test_gpu_fullsize.py tests/test_gpu_sharded.py > $OUT/pytest_orth_reg.log 2>&1; tail -3 $OUT/pytest_orth_reg.log
