OUT=gpurun_out/r02i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_small_gen.py -m gpu -q -s > $OUT/pytest_small_gen.log 2>&1; grep -E "A/B|passed|failed" $OUT/pytest_small_gen.log
timeout 600 python tools/two_process_run.py $OUT/two_process_run.json > $OUT/two_process_run.log 2>&1; head -c 1500 $OUT/two_process_run.json
MISPEC_SMALL_GEN=device timeout 300 python tools/bench_configs.py c4 > $OUT/c4_device_restart.jsonl 2>&1; timeout 300 python tools/bench_configs.py c4 > $OUT/c4_host_restart.jsonl 2>&1; cat $OUT/c4_device_restart.jsonl $OUT/c4_host_restart.jsonl | cut -c1-600
