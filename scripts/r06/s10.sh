#!/bin/bash
# round 6, session 10: where a slow ss_process call's time goes, phase by phase (diagnostics build, ), after another context was closed
OUT=gpurun_out/r06_s10
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp SS_BENCH_SKIP_ALSO_RUNS=1 
SS_BENCH_DROPIN_AT=closed timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --preheat-ms 0 --sets 2 --no-cpu-baseline --no-parity --no-live-pmc > $OUT/trace.out 2> $OUT/trace.err; grep -E "PROBE" $OUT/trace.err | grep -v "65536," | tail -40; tail -5 $OUT/trace.err
