#!/bin/bash
# round 6, GPU call 15: final-tree validation — the whole GPU suite, smoke(), the driver's bench command, its rocprofv3 kernel stats
OUT=gpurun_out/r11o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest -m gpu -q tests > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value', d['value'], 'frac', r['frac'], 'orth_frac', r.get('orth_frac'), 'orth_pass_frac', r.get('orth_pass_frac'), 'csr', r.get('csr_kernel_frac'), 'shard', r.get('shard_proxy_us_per_operation'), r.get('shard_proxy_speedup_8_compute_only'), r.get('shard_proxy_idle_frac_est'), 'turn', r.get('host_turn_us'), 'c5', r.get('secondary_c5_seconds'), 'ref_api', d.get('value_reference_api'))"; tail -3 $OUT/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof; head -6 $OUT/bench_kernel_stats.csv | cut -c1-200
