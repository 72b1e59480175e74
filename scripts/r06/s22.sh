#!/bin/bash
# round 6, session 22: robustness of the final binaries — 500 random detect-mode sessions (culled == unculled; seeds 0..499, 150 of them run
# before), a 5000-step run of the default workload (wait_fallbacks, drains, the rate over 0.1 s), the step-pipeline and stream-order suites thrice
OUT=gpurun_out/r06_s22
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_FUZZ_CULL_SEEDS=500 timeout 2400 python -m pytest tests/test_gpu_cull.py -x -q -m gpu -k random > $OUT/soak500.txt 2>&1; tail -2 $OUT/soak500.txt
timeout 600 python bench.py --gpus 1 --steps 5000 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k5000.json 2>/dev/null
python - <<'PY'
import json
j = json.load(open('/root/repo/bench_full.json'))
print('k5000', j['ms_per_step'], j['value'], j['roofline']['frac'], j['config']['tiles'], j['config']['calls'])
PY
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_step_pipeline.py tests/test_gpu_stream_ordered.py -x -q -m gpu 2>&1 | tail -1; done
