#!/bin/bash
# round 6, GPU call 25: FINAL tree — the whole GPU suite, smoke(), the driver's bench command
OUT=gpurun_out/r11y; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest -m gpu -q tests > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'frac', r['frac'], 'orth_pass_frac', r.get('orth_pass_frac'), 'csr', r.get('csr_kernel_frac'), 'shard', r.get('shard_proxy_us_per_operation'), r.get('shard_proxy_speedup_8_compute_only'), 'turn', r.get('host_turn_us'), 'c4', r.get('secondary_c4_seconds'), 'c5', r.get('secondary_c5_seconds'), 'ref_flow', d.get('value_reference_flow'), 'ref_api', d.get('value_reference_api'), 'csr_value', d.get('value_csr_kernel'))"
