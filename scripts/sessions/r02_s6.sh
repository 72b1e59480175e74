#!/bin/bash
# round 2, GPU session 6: second order sweep (same process, same buffers), role priorities, full GPU suite, bench with preheat
set -x
OUT=gpurun_out/r02_s6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/step_order_lab.py --rounds 3 "E*|D128,F512" "E*|D128,F384" "E*|D128,F768" "E*|D96,F512" "E*|D160,F512" "E*|D256,F512" "E*|D64,F256" "E*,D128|F512,D128" "E*|D128,F1024" "E*|D128,F640" "E*|D192,F768" "|D128,F512,E16" > $OUT/lab_orders2.txt 2>&1; grep round $OUT/lab_orders2.txt
for pr in "1 0" "0 1" "2 1" "1 2" "0 3" "3 0"; do set -- $pr
timeout 300 python scripts/step_order_lab.py --rounds 2 --env SS_STEP_PRIO_FFT=$1 --env SS_STEP_PRIO_OTHER=$2 "E*|D128,F512" "E*|D128,F128" > $OUT/lab_prio_$1_$2.txt 2>&1; echo "PRIO fft=$1 other=$2"; grep round $OUT/lab_prio_$1_$2.txt
done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 300 python bench.py --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20.json 2>&1; tail -c 700 $OUT/bench_20.json
