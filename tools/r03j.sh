OUT=gpurun_out/r03j; mkdir -p $OUT
for cfg in "256,128 32" "256,128 64" "192,128 32" "192,128 64" "160,128 64" "128,128 64"; do set -- $cfg; MISPEC_SHIFT_CHUNK=$1 MISPEC_SHIFT_LANES=$2 timeout 280 python tools/c5_probe.py >> $OUT/c5.jsonl 2>> $OUT/err.log; done
cat $OUT/c5.jsonl; tail -3 $OUT/err.log
