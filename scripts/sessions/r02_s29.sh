#!/bin/bash
# round 2, GPU session 29: ring windows rotating through the whole ring (no wait behind a ring-reading detect stage), eager mode
# for callers that wait after every call; 36 seeds of the long-run test, the whole suite, bench lines
set -x
OUT=gpurun_out/r02_s29; mkdir -p $OUT
SS_TEST_DEEP_SEEDS=36 timeout 900 python -m pytest tests/test_gpu_step_pipeline.py -q -m gpu -k long_runs > $OUT/deep_seeds.txt 2>&1; tail -3 $OUT/deep_seeds.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default_$i.json 2> $OUT/bench_default.err; python -c "import json; d=json.load(open('$OUT/bench_default_$i.json')); print('default', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_k20_$i.json 2> $OUT/bench_k20.err; python -c "import json; d=json.load(open('$OUT/bench_k20_$i.json')); print('k20', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --no-cpu-baseline --sync-every-step > $OUT/bench_sync_$i.json 2> $OUT/bench_sync.err; python -c "import json; d=json.load(open('$OUT/bench_sync_$i.json')); print('sync every step', d['ms_per_step'], d['value'])"
done
