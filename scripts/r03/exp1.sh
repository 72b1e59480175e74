#!/bin/bash
# round 3, experiment 1: tile culling parity + cache-policy A/B + culling on/off (bench.py, 200 steps, no CPU baseline)
OUT=gpurun_out/r03_s1
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_step_pipeline.py tests/test_gpu_stated_configs.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --steps 200 --warmup 5"
for v in base iqnt iqnt_sc0 iqnt_sc1 iqnt_detnt iqnt_psdnt psdnt; do
  timeout 300 $B --lib scripts/ab/libspecscan_$v.so > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  timeout 300 $B --lib scripts/ab/libspecscan_$v.so --no-cull > $OUT/bench_${v}_nocull.json 2> $OUT/bench_${v}_nocull.err
done
timeout 300 $B > $OUT/bench_product.json 2> $OUT/bench_product.err
timeout 300 $B --no-cull > $OUT/bench_product_nocull.json 2> $OUT/bench_product_nocull.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_product_20.json 2> $OUT/bench_product_20.err
timeout 300 $B --no-psd-out > $OUT/bench_product_detect.json 2> $OUT/bench_product_detect.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s1/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
rocprofv3 --list-avail 2>/dev/null | grep -i -E "mall|EA_RDREQ|EA0_RD|DRAM|HBM" | head -40 > $OUT/counters.txt
