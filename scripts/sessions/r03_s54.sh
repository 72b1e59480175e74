#!/bin/bash
# round 3, GPU session 54 (the last): the suite on the final tree (65536-point culling behind SS_CULL_65536 in the diagnostics build), config 3 as shipped
OUT=gpurun_out/r03_s54; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 100 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 40 python bench.py --no-cpu-baseline --config 3 --gpus 1 --preheat-ms 100 > $OUT/bench_cfg3.json 2> $OUT/err
python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r03_s54/bench_cfg3.json').read().splitlines() if l.startswith('{')][-1])
print(j['ms_per_step'], j['value'], {k['slot']: k['us'] for k in j['roofline']['kernels']})
PY
