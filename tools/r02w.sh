OUT=gpurun_out/r02w; mkdir -p $OUT
cp spectra_amd/libmispec.so /tmp/libmispec_default.so
for v in default r13_c16_t512 r14_c15_t256 r14_c15_t512 r14_c15_t1024; do
  if [ $v = default ]; then cp /tmp/libmispec_default.so spectra_amd/libmispec.so; else cp spectra_amd/variants/libmispec_$v.so spectra_amd/libmispec.so; fi
  echo "{\"variant\": \"$v\"}" >> $OUT/mrand.jsonl
  MISPEC_TILES_SYNC=0 timeout 300 python tools/bench_mrand.py 1e7 >> $OUT/mrand.jsonl 2>> $OUT/err.log
done
cp /tmp/libmispec_default.so spectra_amd/libmispec.so
cut -c1-330 $OUT/mrand.jsonl; tail -3 $OUT/err.log
