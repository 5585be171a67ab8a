#!/bin/bash
# round 4, GPU call 8: the whole GPU suite on the tree as it stands, then the driver's bench command with a kernel trace
OUT=gpurun_out/r07h; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
