#!/bin/bash
# round 3, experiment 3: scalar-load tile test
OUT=gpurun_out/r03_s7
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_step_pipeline.py tests/test_gpu_stated_configs.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --steps 200 --warmup 5"
timeout 300 $B > $OUT/bench_product.json 2> $OUT/bench_product.err
timeout 300 $B --no-cull > $OUT/bench_product_nocull.json 2> $OUT/bench_product_nocull.err
timeout 300 $B --start-level 100 > $OUT/bench_product_sl100.json 2> $OUT/bench_product_sl100.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_product_20.json 2> $OUT/bench_product_20.err
timeout 300 $B --no-psd-out > $OUT/bench_product_detect.json 2> $OUT/bench_product_detect.err
timeout 300 $B --lib scripts/ab/libspecscan_base.so > $OUT/bench_abbase.json 2> $OUT/bench_abbase.err
timeout 300 $B --lib scripts/ab/libspecscan_iqnt.so > $OUT/bench_abiqnt.json 2> $OUT/bench_abiqnt.err
SS_CULL_STATS=1 timeout 300 $B --diag-lib > $OUT/bench_diag_stats.json 2> $OUT/bench_diag_stats.err
SS_STEP_STAMPS=$OUT/stamps.txt timeout 300 $B --diag-lib > $OUT/bench_diag_stamps.json 2> $OUT/bench_diag_stamps.err
python scripts/analyze_step_stamps.py $OUT/stamps.txt > $OUT/stamps_summary.txt 2>&1
for o in "E*|D128,F1024" "E*|D*,F*" "E*|D64,F128" "E*|F1024,D*" "D*|E*,F*"; do
  tag=$(echo "$o" | tr -d '*|,' )
  SS_STEP_ORDER="$o" timeout 300 $B --diag-lib > $OUT/bench_order_$tag.json 2> $OUT/bench_order_$tag.err
done
LAB_FFT_ONLY=1 timeout 200 scripts/ubench/launch_overlap_lab > $OUT/lab.txt 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s7/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
grep -h "specscan diag" $OUT/*.err
cat $OUT/stamps_summary.txt; tail -4 $OUT/lab.txt
