#!/bin/bash
# round 6, GPU call 22: the fused V*Q pass through an LDS-DMA ring (vq_dma.hip, option vq_kernel=dma): parity at sizes where it
# applies, then the same-box A/B at C2
OUT=gpurun_out/r11v; mkdir -p $OUT
export TMPDIR=/tmp
MISPEC_VQ_KERNEL=dma timeout 1500 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py tests/test_gpu_fullsize.py -k "benchmark_matrix or c2 or C2 or lds_dma" > $OUT/pytest_vq_dma.log 2>&1; tail -5 $OUT/pytest_vq_dma.log
timeout 1500 python tools/ab_bench.py --steps 3 reg= dma=MISPEC_VQ_KERNEL=dma reg= dma=MISPEC_VQ_KERNEL=dma > $OUT/ab.jsonl 2> $OUT/ab.err; cut -c1-330 $OUT/ab.jsonl; tail -3 $OUT/ab.err
