#!/bin/bash
# round 2, GPU session 7: two-samples-per-lane loads in the FFT role (lab + product), parity of the variants
set -x
OUT=gpurun_out/r02_s7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 scripts/ubench/fft8192_lab 1024 4096 > $OUT/lab.txt 2>&1; grep -v "^#" $OUT/lab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_pipeline.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
for ld in 0 1 2; do
timeout 300 python scripts/step_order_lab.py --rounds 3 --env SS_FFT_LD2=$ld "E*|D128,F1024" "E*|D128,F512" > $OUT/lab_ld2_$ld.txt 2>&1; echo "LD2=$ld"; grep round $OUT/lab_ld2_$ld.txt
done
