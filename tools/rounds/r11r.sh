#!/bin/bash
# round 6, GPU call 18: one rank over the real RCCL transports (the library's communicator, torch.distributed)
OUT=gpurun_out/r11r; mkdir -p $OUT
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_multiproc.py -k "one_rank_over" > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log | cut -c1-400
