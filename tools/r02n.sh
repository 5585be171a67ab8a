OUT=gpurun_out/r02n; mkdir -p $OUT
timeout 900 python tools/ab_bench.py --steps 3 fused= unfused=MISPEC_FUSE_SCALE=0 fused2= unfused2=MISPEC_FUSE_SCALE=0 > $OUT/ab.jsonl 2>&1; cat $OUT/ab.jsonl
