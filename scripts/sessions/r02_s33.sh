#!/bin/bash
# round 2, GPU session 33: deep pipelining with the spectrogram branch on (per-call partial sums, added at the next drain)
set -x
OUT=gpurun_out/r02_s33; mkdir -p $OUT
SS_TEST_DEEP_SEEDS=24 timeout 900 python -m pytest tests/test_gpu_step_pipeline.py -q -m gpu > $OUT/step_pipeline.txt 2>&1; tail -3 $OUT/step_pipeline.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --spectrogram 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spectrogram on', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['value'])"
done
