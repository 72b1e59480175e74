#!/bin/bash
# round 4, session 32: the whole GPU suite on the DIAGNOSTICS build of the final tree (SS_TEST_USE_DIAG_LIB=1: the same sources with
# -DSS_DIAG, no switch set — it must behave like the product), and with the 65536-point chain's older forms switched on suite-wide
OUT=gpurun_out/r04_s32
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_TEST_USE_DIAG_LIB=1 timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_diag.txt 2>&1
echo "diag rc=$?" >> $OUT/rc.txt
SS_TEST_USE_DIAG_LIB=1 SS_DET_LAG2=0 SS_LIST_FIRST=0 timeout 900 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fuzz.py tests/test_gpu_stream_ordered.py tests/test_gpu_degenerate_input.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_lag1.txt 2>&1
echo "lag1 rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; tail -3 $OUT/pytest_diag.txt | cut -c1-300; tail -3 $OUT/pytest_lag1.txt | cut -c1-300
