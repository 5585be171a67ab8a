#!/bin/bash
# round 4, GPU call: bench under rocprofv3 with the live PMC passes left on (a profiler inside a profiler must be skipped or fail softly)
OUT=gpurun_out/r07z; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$OUT/bench_nested.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err; echo "rc=$?")
rm -rf $OUT/prof
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r07z/bench_nested.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:300])
PY
env | grep -i -E "rocp|hsa_tools" | head
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/x -o y -- env | grep -i -E "rocp|hsa_tools|LD_PRELOAD" | head -8)
