OUT=gpurun_out/r02x; mkdir -p $OUT
cp spectra_amd/libmispec.so /tmp/libmispec_default.so
for v in r13_c16_t512 r13_c16_t1024 r12_c17_t512 r12_c17_t1024 r13_c15_t512; do
  cp spectra_amd/variants/libmispec_$v.so spectra_amd/libmispec.so
  echo "{\"variant\": \"$v\"}" >> $OUT/mrand.jsonl
  MISPEC_TILES_SYNC=0 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
done
cp spectra_amd/variants/libmispec_r13_c16_t512.so spectra_amd/libmispec.so
echo "{\"variant\": \"r13_c16_t512 sync 38\"}" >> $OUT/mrand.jsonl
MISPEC_TILES_SYNC=38 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
cp /tmp/libmispec_default.so spectra_amd/libmispec.so
cut -c1-330 $OUT/mrand.jsonl
