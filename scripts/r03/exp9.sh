#!/bin/bash
OUT=gpurun_out/r03_s14
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --warmup 5"
for i in 1 2 3; do
timeout 300 $B --steps 20 > $OUT/bench_product_20_$i.json 2> $OUT/bench_product_20_$i.err
done
timeout 300 $B --steps 200 > $OUT/bench_product_200.json 2> $OUT/bench_product_200.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s14/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['host_enqueue_ms_per_step'], j['config']['tail_us'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
