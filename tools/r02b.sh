OUT=gpurun_out/r02b; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
timeout 900 python tools/ab_bench.py --steps 3 base= dia_diagonal=MISPEC_DIA_LAYOUT=diagonal vq_oop=MISPEC_VQ_OOP=1 base2= > $OUT/ab.jsonl 2>&1; cat $OUT/ab.jsonl
timeout 600 python bench.py --steps 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(json.dumps(d['secondary']['m_rand'], indent=0)[:1800])"; tail -3 $OUT/bench.err
