OUT=gpurun_out/r02r; mkdir -p $OUT
timeout 900 python tools/ab_bench.py --steps 2 base= wg6=MISPEC_DIA_LDS_PAD=16000 wg5=MISPEC_DIA_LDS_PAD=22000 wg4=MISPEC_DIA_LDS_PAD=30000 wg3=MISPEC_DIA_LDS_PAD=43000 base2= > $OUT/ab.jsonl 2>&1; cut -c1-300 $OUT/ab.jsonl
