#!/bin/bash
# round 6, session 11: do streams created after other streams were destroyed copy pageable memory slowly? (scripts/pageable_copy_lab.py)
OUT=gpurun_out/r06_s11
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python scripts/pageable_copy_lab.py > $OUT/pageable_copy_lab.txt 2>&1; tail -8 $OUT/pageable_copy_lab.txt
