#!/bin/bash
OUT=gpurun_out/r03_s11
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_step_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --steps 200 --warmup 5"
timeout 300 $B > $OUT/bench_product.json 2> $OUT/bench_product.err
timeout 300 $B --start-level 100 > $OUT/bench_product_sl100.json 2> $OUT/bench_product_sl100.err
timeout 300 $B --no-cull > $OUT/bench_product_nocull.json 2> $OUT/bench_product_nocull.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_product_20.json 2> $OUT/bench_product_20.err
timeout 300 $B --diag-lib > $OUT/bench_diag.json 2> $OUT/bench_diag.err
for m in 1 2 3; do
SS_ABLATE_ROLES=$m timeout 300 $B --diag-lib > $OUT/bench_diag_ablate$m.json 2> $OUT/bench_diag_ablate$m.err
done
SS_ABLATE_ROLES=3 SS_CULL=0 timeout 300 $B --diag-lib > $OUT/bench_diag_ablate3_nocull.json 2> $OUT/bench_diag_ablate3_nocull.err
timeout 300 $B --lib scripts/ab/libspecscan_base.so > $OUT/bench_abbase.json 2> $OUT/bench_abbase.err
timeout 300 $B --lib scripts/ab/libspecscan_iqnt.so > $OUT/bench_abiqnt.json 2> $OUT/bench_abiqnt.err
timeout 300 $B --lib scripts/ab/libspecscan_base.so > $OUT/bench_abbase2.json 2> $OUT/bench_abbase2.err
timeout 300 $B --lib scripts/ab/libspecscan_iqnt.so > $OUT/bench_abiqnt2.json 2> $OUT/bench_abiqnt2.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s11/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
