OUT=gpurun_out/r02y; mkdir -p $OUT
cp spectra_amd/libmispec.so /tmp/libmispec_default.so
for v in r13_c17_t512_k2048 r13_c17_t1024_k2048 r14_c16_t1024_k2048 r13_c16_t512_k1024 r13_c17_t512_k1024; do
  cp spectra_amd/variants/libmispec_$v.so spectra_amd/libmispec.so
  echo "{\"variant\": \"$v\"}" >> $OUT/mrand.jsonl
  MISPEC_TILES_SYNC=0 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
done
cp /tmp/libmispec_default.so spectra_amd/libmispec.so
cut -c1-330 $OUT/mrand.jsonl
