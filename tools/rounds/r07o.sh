#!/bin/bash
# round 4, GPU call: the driver's bench command with the live PMC passes (roofline.traffic measured in the run)
OUT=gpurun_out/r07o; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench wall seconds: $(( $(date +%s) - T0 ))"

python - <<'PY'
import json
d = json.loads(open("gpurun_out/r07o/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic"], r.get("traffic_committed"))
print(r["traffic_source"][:400])
print(d["secondary"]["c5"] if "secondary" in d else None)
PY
