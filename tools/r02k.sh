OUT=gpurun_out/r02k; mkdir -p $OUT
timeout 900 python tools/bench_stencil.py 215 > $OUT/stencil.jsonl 2> $OUT/stencil.err; cat $OUT/stencil.jsonl; tail -3 $OUT/stencil.err
for s in 16 24 30; do MISPEC_TILES_SYNC=$s timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err; done; cat $OUT/mrand.jsonl
