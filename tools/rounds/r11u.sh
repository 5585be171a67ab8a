#!/bin/bash
# round 6, GPU call 21: the whole GPU suite + smoke on the tree with orth_dma_modes.hip
OUT=gpurun_out/r11u; mkdir -p $OUT
timeout 1800 python -m pytest -m gpu -q tests > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
