OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 500 python tools/ab_bench.py --steps 2 base= rev=MISPEC_ORTH_REVERSE=1 base2= rev2=MISPEC_ORTH_REVERSE=1 > $OUT/ab_orth_reverse.jsonl 2> $OUT/ab.err; cut -c1-420 $OUT/ab_orth_reverse.jsonl
