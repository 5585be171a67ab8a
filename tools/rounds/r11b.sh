#!/bin/bash
# round 6, GPU call 2: the LDS-DMA one-sweep pass (orth_dma.hip): parity (one-sweep module, C2 golden) and A/B on C2
OUT=gpurun_out/r11b; mkdir -p $OUT
export TMPDIR=/tmp
MISPEC_ORTH_KERNEL=dma timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py > $OUT/pytest_onesweep_dma.log 2>&1; tail -5 $OUT/pytest_onesweep_dma.log
MISPEC_ORTH_KERNEL=dma2 timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py -k "benchmark_matrix or fixtures" > $OUT/pytest_onesweep_dma2.log 2>&1; tail -3 $OUT/pytest_onesweep_dma2.log
timeout 1200 python tools/ab_bench.py --steps 3 reg= dma=MISPEC_ORTH_KERNEL=dma dma2=MISPEC_ORTH_KERNEL=dma2 reg= dma=MISPEC_ORTH_KERNEL=dma > $OUT/ab.jsonl 2> $OUT/ab.err; cut -c1-700 $OUT/ab.jsonl; tail -3 $OUT/ab.err
