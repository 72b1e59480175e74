#!/bin/bash
# round 2, GPU session 13: final-state validation — full GPU suite, extended seeds, smoke, the bench line
set -x
OUT=gpurun_out/r02_s13; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
SS_FUZZ_SEEDS=120 SS_FUZZ_SEEDS2=80 SS_FUZZ_SEEDS3=40 SS_FUZZ_SEEDS4=40 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_fuzz_long.txt 2>&1; tail -4 $OUT/pytest_fuzz_long.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1800 $OUT/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2>&1; tail -c 500 $OUT/bench20.json
