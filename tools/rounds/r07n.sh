#!/bin/bash
# round 4, GPU call: staged / tile formats on row shards
OUT=gpurun_out/r07n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_sharded.py tests/test_gpu_staged.py tests/test_gpu_tiles.py tests/test_gpu_multiproc.py > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
