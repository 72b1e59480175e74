#!/bin/bash
# round 4, session 31: two and three independent 65536-point bands on one GPU (one context, stream and host thread each): what would
# launches of two queues side by side be worth at this size today?
OUT=gpurun_out/r04_s31
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python scripts/multi_band_rate.py --fft 65536 --frames 128 --steps 200 --cs8 --detect --bands 1 2 3 > $OUT/bands_f128.txt 2>&1
timeout 300 python scripts/multi_band_rate.py --fft 65536 --frames 64 --steps 300 --cs8 --detect --bands 1 2 4 > $OUT/bands_f64.txt 2>&1
timeout 300 python scripts/multi_band_rate.py --fft 1048576 --frames 16 --steps 60 --detect --bands 1 2 > $OUT/bands_2p20.txt 2>&1
grep -h bands_on $OUT/bands_*.txt | cut -c1-200; tail -2 $OUT/bands_2p20.txt | cut -c1-200
