#!/bin/bash
# round 6, GPU call 6: the whole GPU suite on the tree of session 1, then the clean evidence VERDICT r05 item 3 asks for:
# rocprofv3 --kernel-trace --stats of C2-ONLY solves (headline k_spmv_dia_win2; forced int32 CSR k_spmv_csr_win), and the
# FETCH_SIZE / WRITE_SIZE passes of the probe for both formats (one-sweep flow only).
OUT=gpurun_out/r11f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest -m gpu -q -x tests > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
for F in dia:-1 csr:0; do
  NAME=${F%%:*}; FMT=${F##*:}
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$NAME -o t -- python $GRAFT_REPO_ROOT/tools/c2_solves.py --format $FMT --solves 3 > $GRAFT_REPO_ROOT/$OUT/c2_only_${NAME}_stdout.json 2> $GRAFT_REPO_ROOT/$OUT/c2_only_${NAME}.err)
  find $OUT/prof_$NAME -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c2_only_${NAME}_kernel_stats.csv
  rm -rf $OUT/prof_$NAME; head -5 $OUT/c2_only_${NAME}_kernel_stats.csv | cut -c1-220; tail -1 $OUT/c2_only_${NAME}_stdout.json | cut -c1-400
done
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && PROBE_FORMATS=2,0 PROBE_MODES=onesweep timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py > $GRAFT_REPO_ROOT/$OUT/pmc_$C.log 2>&1)
  find $OUT/pmc_$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc_${C}_counter_collection.csv
  rm -rf $OUT/pmc_$C
done
python tools/pmc_summarize.py $OUT/pmc_FETCH_SIZE_counter_collection.csv $OUT/pmc_WRITE_SIZE_counter_collection.csv 10000000 $OUT/c2_pmc_traffic.json > $OUT/pmc_summary.txt 2>&1; tail -25 $OUT/pmc_summary.txt | cut -c1-200
rm -f $OUT/pmc_*_counter_collection.csv
