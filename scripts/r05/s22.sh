#!/bin/bash
# round 5, session 22: soak — sixty random detect-mode sessions (sizes 8192 ... 2^20 incl. 131072, int8 and CF32, call sizes from one
# frame up, learning inside or across calls, retunes with and without a reset, zero-frame calls, host and device entry points mixed):
# the culled forms (the fold, dB ring rows, the form changes between them) against SS_FLAG_NO_CULL's (round 2's paths), list by list
OUT=gpurun_out/r05_s22
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_FUZZ_CULL_SEEDS=60 timeout 2400 python -m pytest tests/test_gpu_cull.py -m gpu -q -k random_detect_mode > $OUT/soak.txt 2>&1
tail -5 $OUT/soak.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q > $OUT/fuzz_parity.txt 2>&1
tail -3 $OUT/fuzz_parity.txt | cut -c1-300
