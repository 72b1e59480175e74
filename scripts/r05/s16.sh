#!/bin/bash
# round 5, session 16: the whole GPU suite with the radix-16 fold in the tree, once on the product library and once on the diagnostics twin
OUT=gpurun_out/r05_s16
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt | cut -c1-300
SS_TEST_USE_DIAG_LIB=1 timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_diag.txt 2>&1
tail -6 $OUT/pytest_gpu_diag.txt | cut -c1-300
