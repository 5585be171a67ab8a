OUT=gpurun_out/r02l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fac.py tests/test_gpu_solver.py -m gpu -q -x -k "not full_size" > $OUT/pytest_direct.log 2>&1 &
wait
MISPEC_VQ=direct timeout 600 python -m pytest tests/test_gpu_fac.py tests/test_gpu_solver.py -m gpu -q -k "not full_size and not device_driven and not ritz_pairs" > $OUT/pytest_direct.log 2>&1; tail -3 $OUT/pytest_direct.log
timeout 900 python tools/ab_bench.py --steps 3 base= vq_direct=MISPEC_VQ=direct vq_direct2=MISPEC_VQ=direct,MISPEC_VQ_BLOCKS_PER_CU=2 vq_direct8=MISPEC_VQ=direct,MISPEC_VQ_BLOCKS_PER_CU=8 base2= > $OUT/ab.jsonl 2>&1; cat $OUT/ab.jsonl
