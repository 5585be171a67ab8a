#!/bin/bash
# round 6, GPU call 19: bench.py with no flags (the contract's default) — wall time and the line
OUT=gpurun_out/r11s; mkdir -p $OUT
export TMPDIR=/tmp
S=$(date +%s); python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "wall seconds: $(( $(date +%s) - S ))" | tee $OUT/bench_default_wall.txt
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print('roofline', {k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic')}, 'cpu_baseline', {k:d['cpu_baseline'][k] for k in ('value','unit','cores','kind')})"
