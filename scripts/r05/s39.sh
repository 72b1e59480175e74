#!/bin/bash
# round 5, session 39: soak of the final binaries — 150 random detect-mode sessions (sizes 8192 ... 2^20 incl. 131072, int8 and CF32, call
# sizes from one frame up, learning inside or across calls, retunes with and without a reset, zero-frame calls, host and device entry
# points mixed): the culled forms (halo maxima, the fold, dB ring rows, the general path on lanes) against SS_FLAG_NO_CULL's, list by list
OUT=gpurun_out/r05_s39
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_FUZZ_CULL_SEEDS=150 timeout 600 python -m pytest tests/test_gpu_cull.py -m gpu -q -k random_detect_mode > $OUT/soak.txt 2>&1
tail -3 $OUT/soak.txt | cut -c1-300
