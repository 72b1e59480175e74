#!/bin/bash
OUT=gpurun_out/r03_tests
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 90 python bench.py --no-cpu-baseline --steps 200 --warmup 5 > $OUT/bench_product.json 2> $OUT/bench_product.err
timeout 90 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_product_20.json 2> $OUT/bench_product_20.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_tests/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
