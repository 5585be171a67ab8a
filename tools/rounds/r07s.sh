#!/bin/bash
# round 4, GPU call: Infinity-Cache probe for a group-wise staged product; staged tests + M-rand with two chunks per wavefront
OUT=gpurun_out/r07s; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 tools/probes/mall_probe.bin > $OUT/mall_probe.jsonl 2> $OUT/mall_probe.err; cat $OUT/mall_probe.jsonl; tail -2 $OUT/mall_probe.err
timeout 600 python -m pytest -m gpu -q -x tests/test_gpu_staged.py tests/test_gpu_sharded.py -k "staged or scattered" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
BENCH_FORMATS=4 timeout 600 python tools/bench_staged.py > $OUT/mrand_staged.jsonl 2> $OUT/mrand_staged.err; tail -1 $OUT/mrand_staged.jsonl | cut -c1-300
