OUT=gpurun_out/r02j; mkdir -p $OUT
for s in 1000 77 39 20 10; do MISPEC_TILES_SYNC=$s timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/mrand.err; done
cat $OUT/mrand.jsonl
MISPEC_TILES_SYNC=77 bash tools/pmc_pass.sh $OUT l2_sync77 "TCC_HIT_sum TCC_MISS_sum" tools/pmc_probe_mrand.py
