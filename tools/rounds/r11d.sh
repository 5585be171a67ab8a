#!/bin/bash
# round 6, GPU call 4: host turn through pinned memory (option host_turn), one-round record reduction, LDS-DMA pass variants:
# parity, then same-box A/Bs at shard size (1.25 M rows) and at C2, then the shard-size kernel trace with its gaps
OUT=gpurun_out/r11d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py tests/test_gpu_fac.py tests/test_gpu_solver.py > $OUT/pytest_default.log 2>&1; tail -4 $OUT/pytest_default.log
MISPEC_ORTH_KERNEL=dma timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py tests/test_gpu_sharded.py > $OUT/pytest_dma.log 2>&1; tail -3 $OUT/pytest_dma.log
MISPEC_HOST_TURN=copy timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py -k "restart or fixtures" > $OUT/pytest_copy.log 2>&1; tail -3 $OUT/pytest_copy.log
for rep in 1 2; do
 for v in "copy reg" "fast reg" "fast dma"; do
  set -- $v
  MISPEC_HOST_TURN=$1 MISPEC_ORTH_KERNEL=$2 python tools/shard_profile.py 1250000 onesweep 4 2>>$OUT/shard.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'v':'$v','s_per_solve':round(d['seconds_per_solve'],5),'us_per_op':round(1e3*d['ms_per_operator_application_all_inclusive'],2),'nops':d['num_operations'],'turn':d['turn_info'],'fam':{k:round(v,4) for k,v in d['kernel_families_ms_per_operation'].items()}}))" | tee -a $OUT/shard_ab.jsonl
 done
done
(cd /tmp && MISPEC_HOST_TURN=fast MISPEC_ORTH_KERNEL=dma timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/shard_profile.py 1250000 onesweep 3 > $GRAFT_REPO_ROOT/$OUT/trace_stdout.txt 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_1250000_rows.csv
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T k_orth_lagged > $OUT/trace_gaps_1250000_rows.txt; head -12 $OUT/trace_gaps_1250000_rows.txt
rm -rf $OUT/prof; head -8 $OUT/kernel_stats_1250000_rows.csv | cut -c1-200
timeout 1500 python tools/ab_bench.py --steps 3 copy_reg=MISPEC_HOST_TURN=copy,MISPEC_ORTH_KERNEL=reg fast_reg=MISPEC_ORTH_KERNEL=reg fast_dma=MISPEC_ORTH_KERNEL=dma fast_dmac=MISPEC_ORTH_KERNEL=dmac fast_dmap=MISPEC_ORTH_KERNEL=dmap fast_dma=MISPEC_ORTH_KERNEL=dma copy_reg=MISPEC_HOST_TURN=copy,MISPEC_ORTH_KERNEL=reg > $OUT/ab.jsonl 2> $OUT/ab.err; cut -c1-420 $OUT/ab.jsonl; tail -3 $OUT/ab.err
