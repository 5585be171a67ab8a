#!/bin/bash
# round 4, GPU call: latency of small solves; host-eigenvector figure after the huge-page hint
OUT=gpurun_out/r07t; mkdir -p $OUT
export TMPDIR=/tmp
cat /sys/kernel/mm/transparent_hugepage/enabled > $OUT/thp.txt 2>&1
timeout 600 python tools/small_latency.py > $OUT/small_latency.jsonl 2> $OUT/small_latency.err; cat $OUT/small_latency.jsonl | cut -c1-420; tail -3 $OUT/small_latency.err
timeout 600 python bench.py --steps 3 --no-cpu-baseline --no-secondary --no-live-pmc > $OUT/bench_short.json 2> $OUT/bench_short.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r07t/bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["value_with_host_eigenvectors"])
PY
cat $OUT/thp.txt
