#!/bin/bash
# round 4, GPU call: staged format with row|rank stored at the phase-1 positions (no per-chunk padding)
OUT=gpurun_out/r07p; mkdir -p $OUT
export TMPDIR=/tmp
echo tests ran in the previous call
BENCH_FORMATS=4 timeout 600 python tools/bench_staged.py > $OUT/mrand_staged.jsonl 2> $OUT/mrand_staged.err; cut -c1-900 $OUT/mrand_staged.jsonl; tail -3 $OUT/mrand_staged.err
