#!/bin/bash
# round 6, GPU call 3: counters of the LDS-DMA pass (same probe as r11a) + kernel trace of C2 solves with it
OUT=gpurun_out/r11c; mkdir -p $OUT
export TMPDIR=/tmp
export MISPEC_ORTH_KERNEL=dma
PROBE_FORMATS=2 PROBE_MODES=onesweep PROBE_SPMV_REPS=1 PMC_GROUPS=sq,sq2,tcc,tcp,grbm python tools/pmc_counters.py collect $OUT -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py
python tools/pmc_counters.py summarize $OUT k_orth_lagged k_reduce > $OUT/summary.txt
awk '/k_orth_lagged_dma<10/,/read latency/' $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o t -- python $GRAFT_REPO_ROOT/tools/ab_orth.py --reps 1 --solves 2 onesweep > $GRAFT_REPO_ROOT/$OUT/trace_stdout.txt 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/prof; head -12 $OUT/kernel_stats.csv | cut -c1-200; tail -2 $OUT/trace_stdout.txt | cut -c1-300
