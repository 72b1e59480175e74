#!/bin/bash
# round 6, session 10: what in bench.py's process slows ss_process's 128 MiB call down tenfold (bisect)
OUT=gpurun_out/r06_s10
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 600 python scripts/drop_in_trace.py --bisect > $OUT/drop_in_bisect.txt 2>&1; cat $OUT/drop_in_bisect.txt
