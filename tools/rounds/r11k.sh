#!/bin/bash
# round 6, GPU call 11: the LDS-DMA / register pass equivalence test, options test, smoke(); gap percentiles at shard size
OUT=gpurun_out/r11k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py -k "lds_dma or fused_restart or benchmark_matrix" > $OUT/pytest_dma_equivalence.log 2>&1; tail -4 $OUT/pytest_dma_equivalence.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/c2_solves.py --size 1250000 --solves 5 > $GRAFT_REPO_ROOT/$OUT/shard_only_stdout.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $T k_orth_lagged > $OUT/trace_gaps_1250000_rows_percentiles.txt; head -14 $OUT/trace_gaps_1250000_rows_percentiles.txt | cut -c1-175
python - <<'PY' $T > $OUT/gap_positions.txt
import csv, sys
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# gaps in front of k_orth_lagged_dma<9, ...>: which launch of its sweep, how long
big=[]
for (s0,e0,n0),(s1,e1,n1) in zip(rows, rows[1:]):
    if "k_orth_lagged_dma<9" in n1 and "k_reduce_partials" in n0:
        big.append((s1-e0)/1e3)
big.sort()
print("count", len(big), "median", big[len(big)//2], "p75", big[3*len(big)//4], "p90", big[9*len(big)//10], "max", big[-1])
print("over 10 us:", sum(g>10 for g in big), "sum us of those", sum(g for g in big if g>10))
PY
cat $OUT/gap_positions.txt
rm -rf $OUT/prof
