#!/bin/bash
OUT=gpurun_out/r03_s20
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --steps 200 --warmup 5"
timeout 90 $B > $OUT/bench_product.json 2> $OUT/bench_product.err
timeout 90 $B --start-level 100 > $OUT/bench_product_sl100.json 2> $OUT/bench_product_sl100.err
timeout 90 $B --no-cull > $OUT/bench_product_nocull.json 2> $OUT/bench_product_nocull.err
timeout 90 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_product_20.json 2> $OUT/bench_product_20.err
timeout 90 $B --no-psd-out > $OUT/bench_product_detect.json 2> $OUT/bench_product_detect.err
timeout 90 $B --fmt cs8 > $OUT/bench_product_cs8.json 2> $OUT/bench_product_cs8.err
timeout 90 $B --spectrogram > $OUT/bench_product_spec.json 2> $OUT/bench_product_spec.err
timeout 90 $B --planes > $OUT/bench_product_planes.json 2> $OUT/bench_product_planes.err
timeout 90 $B --frames 256 > $OUT/bench_product_f256.json 2> $OUT/bench_product_f256.err
timeout 90 $B --frames 4096 > $OUT/bench_product_f4096.json 2> $OUT/bench_product_f4096.err
SS_STEP_STAMPS=$OUT/stamps.txt timeout 90 $B --diag-lib > $OUT/bench_diag_stamps.json 2> $OUT/bench_diag_stamps.err
python scripts/analyze_step_stamps.py $OUT/stamps.txt 2>/dev/null | head -12 > $OUT/stamps_summary.txt
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s20/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/stamps_summary.txt
