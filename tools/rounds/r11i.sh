#!/bin/bash
# round 6, GPU call 9: two libraries (libmispec.so + libmispec_extras.so): the whole GPU suite; set_shift phase profile of C5 and
# the wide-band case W5 (host levels factored by threads); bench line with the pass-only orth figure (profile level 4)
OUT=gpurun_out/r11i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest -m gpu -q -x tests > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
MISPEC_SHIFT=profile=1 python tools/bench_configs.py c5 > $OUT/c5.json 2> $OUT/c5_set_shift_phases.txt; cat $OUT/c5_set_shift_phases.txt | head -40; cut -c1-700 $OUT/c5.json
MISPEC_SHIFT=profile=1 python tools/bench_configs.py w5 > $OUT/w5.json 2> $OUT/w5_set_shift_phases.txt; tail -12 $OUT/w5_set_shift_phases.txt; cut -c1-700 $OUT/w5.json
W5_B=12 MISPEC_SHIFT=profile=1 python tools/bench_configs.py w5 > $OUT/w5_b12.json 2> $OUT/w5_b12_set_shift_phases.txt; tail -8 $OUT/w5_b12_set_shift_phases.txt; cut -c1-500 $OUT/w5_b12.json
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-pmc --no-secondary > $OUT/bench_quick.json 2> $OUT/bench_quick.err; python -c "
import json; d=json.loads(open('$OUT/bench_quick.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline'].get('orth_frac'), d['roofline'].get('orth_pass_frac'), d['roofline_orth'].get('pass_only'))"; tail -3 $OUT/bench_quick.err
