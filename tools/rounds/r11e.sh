#!/bin/bash
# round 6, GPU call 5: preloaded scalar tail of the record reduction; unsharded shard-size profile (bench.py --size 1250000)
OUT=gpurun_out/r11e; mkdir -p $OUT
export TMPDIR=/tmp
export MISPEC_ORTH_KERNEL=dma
timeout 1200 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py tests/test_gpu_fac.py tests/test_gpu_sharded.py > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --size 1250000 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-live-pmc --no-profile > $GRAFT_REPO_ROOT/$OUT/bench_1250000.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_1250000_rows.csv
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T k_orth_lagged > $OUT/trace_gaps_1250000_rows.txt; head -14 $OUT/trace_gaps_1250000_rows.txt
rm -rf $OUT/prof; head -9 $OUT/kernel_stats_1250000_rows.csv | cut -c1-220
python bench.py --size 1250000 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --no-live-pmc > $OUT/bench_1250000_plain.json 2>$OUT/bench_plain.err; python -c "
import json; d=json.loads(open('$OUT/bench_1250000_plain.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['solve']['num_operations'], d['solve']['host_syncs_per_solve'], d['kernels_ms_per_solve'])"
