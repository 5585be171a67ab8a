#!/bin/bash
# round 6, GPU call 20: the reference-flow / Arnoldi passes through the LDS ring (orth_dma_modes.hip): equivalence, the modules
# that run those flows, C4 and the reference flow timed with and without
OUT=gpurun_out/r11t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest -m gpu -q -x tests/test_gpu_onesweep.py -k "reference_flow_equal" > $OUT/pytest_equiv.log 2>&1; tail -6 $OUT/pytest_equiv.log
MISPEC_ORTH=reference timeout 1700 python -m pytest -m gpu -q -x tests/test_gpu_fullsize.py tests/test_gpu_gen.py tests/test_gpu_fac.py tests/test_gpu_sharded.py tests/test_gpu_svd.py > $OUT/pytest_reference_flow.log 2>&1; tail -5 $OUT/pytest_reference_flow.log
for v in reg dma reg dma; do
  if [ $v = reg ]; then export MISPEC_ORTH_KERNEL=reg; else unset MISPEC_ORTH_KERNEL; fi
  python tools/bench_configs.py c4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'v':'$v','c4_seconds':round(d['seconds'],4),'nops':d['num_operations'],'kernels_ms':d['kernels_ms']}))" | tee -a $OUT/c4_ab.jsonl
  python tools/c2_solves.py --orth reference --solves 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'v':'$v','reference_flow_eigenpairs_per_s':round(d['eigenpairs_per_s'],3),'nops':d['num_operations'],'niter':d['num_iterations']}))" | tee -a $OUT/reference_flow_ab.jsonl
done
