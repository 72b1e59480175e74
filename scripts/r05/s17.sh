#!/bin/bash
# round 5, session 17: how much of the fold's kernel is waiting for its LDS-DMA pieces — the lab with the fetches and their waits taken
# out (garbage results: arithmetic, LDS reads and barriers only) against the kernel as it is
OUT=gpurun_out/r05_s17
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
DIF8_ONLY=4 timeout 300 scripts/ubench/dif8_lab 128 512 > $OUT/lab_v4.txt 2>&1
DIF8_ONLY=4 timeout 300 scripts/ubench/dif8_lab_nodma 128 512 > $OUT/lab_v4_nodma.txt 2>&1
DIF8_ONLY=1 timeout 300 scripts/ubench/dif8_lab 128 512 > $OUT/lab_v1.txt 2>&1
DIF8_ONLY=1 timeout 300 scripts/ubench/dif8_lab_nodma 128 512 > $OUT/lab_v1_nodma.txt 2>&1
grep frames $OUT/lab_v4.txt $OUT/lab_v4_nodma.txt $OUT/lab_v1.txt $OUT/lab_v1_nodma.txt | cut -c1-220
