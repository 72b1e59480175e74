#!/bin/bash
# round 2, GPU session 9: FFT workgroups that take several frames (two FFT slots per CU)
set -x
OUT=gpurun_out/r02_s9; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python scripts/step_order_lab.py --rounds 3 "F*,E*|D*" "E*,D128,F*|D*" "E*,F*|D*" "F*|D*,E*" "F*,D128,E*|D*" "E*,D256,F*|D*" "D256,F*,E*|D*" > $OUT/lab_auto.txt 2>&1; grep round $OUT/lab_auto.txt
timeout 300 python scripts/step_order_lab.py --rounds 2 --env SS_FFT_PER_WG=1 "E*|D128,F1024" "F*,E*|D*" > $OUT/lab_per1.txt 2>&1; echo PER_WG=1; grep round $OUT/lab_per1.txt
timeout 300 python scripts/step_order_lab.py --rounds 2 --env SS_FFT_PER_WG=4 "F*,E*|D*" "E*,D128,F*|D*" > $OUT/lab_per4.txt 2>&1; echo PER_WG=4; grep round $OUT/lab_per4.txt
timeout 300 python scripts/step_order_lab.py --rounds 2 --frames 2048 "F*,E*|D*" "E*,D128,F*|D*" > $OUT/lab_2048.txt 2>&1; echo FRAMES=2048; grep round $OUT/lab_2048.txt
timeout 300 python scripts/step_order_lab.py --rounds 2 --frames 512 "F*,E*|D*" "E*,D128,F*|D*" > $OUT/lab_512.txt 2>&1; echo FRAMES=512; grep round $OUT/lab_512.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_pipeline.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
