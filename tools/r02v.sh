OUT=gpurun_out/r02v; mkdir -p $OUT
timeout 300 python tools/bench_mrand.py 1e7 > $OUT/mrand.jsonl 2> $OUT/err.log
MISPEC_TILES_SYNC=0 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
cat $OUT/mrand.jsonl; tail -2 $OUT/err.log
