#!/bin/bash
# round 4, GPU call: hipGraph probe (launch-bound small solves), then the whole GPU suite on the current tree
OUT=gpurun_out/r07u; mkdir -p $OUT
export TMPDIR=/tmp
for n in 1000 100000; do timeout 120 tools/probes/graph_probe.bin 150 $n; done > $OUT/graph_probe.jsonl 2> $OUT/graph_probe.err; cat $OUT/graph_probe.jsonl; tail -2 $OUT/graph_probe.err
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
