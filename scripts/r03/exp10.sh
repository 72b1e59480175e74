#!/bin/bash
OUT=gpurun_out/r03_s15
mkdir -p $OUT
cd /root/repo
for q in 3 4; do
SS_TEST_USE_DIAG_LIB=1 SS_QUEUES=$q timeout 900 python -m pytest tests/test_gpu_cull.py tests/test_gpu_step_pipeline.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest_q$q.log 2>&1
echo "pytest q=$q rc=$?" >> $OUT/pytest_q$q.log
tail -3 $OUT/pytest_q$q.log
done
B="python bench.py --no-cpu-baseline --warmup 5 --diag-lib"
for q in 2 3 4; do
SS_QUEUES=$q timeout 300 $B --steps 200 > $OUT/bench_q${q}_200.json 2> $OUT/bench_q${q}_200.err
SS_QUEUES=$q timeout 300 $B --steps 20 > $OUT/bench_q${q}_20.json 2> $OUT/bench_q${q}_20.err
SS_QUEUES=$q timeout 300 $B --steps 200 --no-cull > $OUT/bench_q${q}_200_nocull.json 2> $OUT/bench_q${q}_200_nocull.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s15/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['host_enqueue_ms_per_step'], j['config']['tail_us'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
