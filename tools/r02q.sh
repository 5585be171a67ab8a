OUT=gpurun_out/r02q; mkdir -p $OUT
timeout 900 python tools/ab_bench.py --steps 2 int32=MISPEC_SPMV_DIA=0,MISPEC_SPMV_CODES=0 int32_small=MISPEC_SPMV_DIA=0,MISPEC_SPMV_CODES=0,MISPEC_SPMV_SMALL_CHUNK=1 int32_b=MISPEC_SPMV_DIA=0,MISPEC_SPMV_CODES=0 > $OUT/ab.jsonl 2>&1; cut -c1-330 $OUT/ab.jsonl
