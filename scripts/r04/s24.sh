#!/bin/bash
# round 4, session 24: a soak of the culling paths as they ship (65536 points: two detect stages waiting, the ring wrapping under them):
# 60 random detect-mode sessions (24 of them at 65536 points, 12 at 2^20), culled == unculled list by list
OUT=gpurun_out/r04_s24
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_FUZZ_CULL_SEEDS=60 timeout 1200 python -m pytest tests/test_gpu_cull.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k random > $OUT/pytest_soak.txt 2>&1
echo "soak rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; tail -5 $OUT/pytest_soak.txt | cut -c1-400
