#!/bin/bash
# round 5, session 14: the fold with radix 16 — 131072-point frames (what getFft picks at 20 MS/s), two residues per workgroup — in the lab
OUT=gpurun_out/r05_s14
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
DIF_Q=16 timeout 300 scripts/ubench/dif8_lab 64 128 256 32 > $OUT/dif16_lab.txt 2>&1
DIF8_ONLY=4 timeout 300 scripts/ubench/dif8_lab 128 > $OUT/dif8_lab_v4.txt 2>&1
cat $OUT/dif16_lab.txt; tail -3 $OUT/dif8_lab_v4.txt
