#!/bin/bash
# round 4, GPU call 1: int32 CSR kernel in the loop by flow and chunk size; the whole GPU suite with one-sweep as the default
OUT=gpurun_out/r07a; mkdir -p $OUT
for K in 2 4; do MISPEC_CSR_CHUNK=$K timeout 300 python tools/probe_csr_variants.py 0 >> $OUT/csr_variants.jsonl 2>> $OUT/csr_variants.err; done
timeout 200 python tools/probe_csr_variants.py 1 >> $OUT/csr_variants.jsonl 2>> $OUT/csr_variants.err
cat $OUT/csr_variants.jsonl
MISPEC_ORTH=onesweep timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_onesweep_default.log 2>&1
tail -40 $OUT/pytest_onesweep_default.log
