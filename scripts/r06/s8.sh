#!/bin/bash
# round 6, session 8: where ss_process spends 24 ms of a 2^20-point call inside bench.py's drop_in_lines (alone: 2.7 ms = 6.2 GS/s; the copy
# takes 2.4 ms by every API: s7's lab) — drop_in_lines in a process of its own, plain and under rocprofv3 --hip-trace --stats
OUT=gpurun_out/r06_s8
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python scripts/drop_in_trace.py > $OUT/drop_in_plain.txt 2>&1; cat $OUT/drop_in_plain.txt
timeout 300 python -c "
import sys, json; sys.argv=['bench.py']
import bench
for e in bench.drop_in_lines(): print({k: e.get(k) for k in ('fft_size','ss_process_MSps','ss_feed_MSps','ss_process_pieces_ms','error')})
" > $OUT/drop_in_lines_alone.txt 2>&1; cat $OUT/drop_in_lines_alone.txt
cd /tmp
PYTHONPATH=/root/repo timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/dtrace -- python -c "
import sys, json; sys.argv=['bench.py']; sys.path.insert(0, '/root/repo')
import os; os.chdir('/root/repo')
import bench
for e in bench.drop_in_lines(): print({k: e.get(k) for k in ('fft_size','ss_process_MSps','ss_feed_MSps','ss_process_pieces_ms','error')})
" > /root/repo/$OUT/drop_in_traced.txt 2>&1
cd /root/repo
grep fft_size $OUT/drop_in_traced.txt
S=$(find /tmp/dtrace -name "*hip_api_stats.csv" | head -1); cp $S $OUT/hip_api_stats.csv 2>/dev/null; head -16 $OUT/hip_api_stats.csv | cut -c1-160
