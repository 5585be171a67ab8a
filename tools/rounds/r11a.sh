#!/bin/bash
# round 6, GPU call 1: SQ / TCC / TCP counter breakdown of the kernels of a one-sweep C2 factorisation on the tree of round 5
# (k_orth_lagged first: VERDICT r05 item 2), diagonal storage, one flow per pass.
OUT=gpurun_out/r11a; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters_available.txt 2>&1)
grep -c . $OUT/counters_available.txt
PROBE_FORMATS=2 PROBE_MODES=onesweep PROBE_SPMV_REPS=1 python tools/pmc_counters.py collect $OUT -- python $GRAFT_REPO_ROOT/tools/pmc_probe.py
python tools/pmc_counters.py summarize $OUT k_orth_lagged k_spmv k_vq k_reduce > $OUT/summary.txt
head -150 $OUT/summary.txt
