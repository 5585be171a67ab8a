#!/bin/bash
# round 6, GPU call 7: the restart's QR sweeps as a skewed pipeline — device kernel against the host routine (bit for bit), the
# latency of every variant, the solver tests with the pipelined host routine as the default, and A/Bs of the host turn.
OUT=gpurun_out/r11g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_small.py > $OUT/pytest_small.log 2>&1; tail -4 $OUT/pytest_small.log
for a in "40 18" "40 26" "64 40" "24 12"; do python tools/restart_sweeps_latency.py $a | tee -a $OUT/restart_sweeps_latency.jsonl | cut -c1-600; done
timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py tests/test_gpu_fac.py tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py > $OUT/pytest_solver.log 2>&1; tail -4 $OUT/pytest_solver.log
for rep in 1 2; do
 for v in host-serial host; do
  MISPEC_SMALL=$v python bench.py --size 1250000 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --no-live-pmc 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'small':'$v','value':round(d['value'],2),'ms_per_solve':round(d['ms_per_step'],3),'nops':d['solve']['num_operations'],'host_turn_us':d['solve'].get('host_turn_us')}))" | tee -a $OUT/shard_size_ab.jsonl
 done
done
timeout 1200 python tools/ab_bench.py --steps 3 serial=MISPEC_SMALL=host-serial pipelined= serial=MISPEC_SMALL=host-serial pipelined= > $OUT/ab_c2.jsonl 2> $OUT/ab_c2.err; cut -c1-330 $OUT/ab_c2.jsonl; tail -3 $OUT/ab_c2.err
