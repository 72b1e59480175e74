#!/bin/bash
# round 2, GPU session 1: FFT lab on an HBM-sized working set + the reworked bench.py
set -x
OUT=gpurun_out/r02_s1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 scripts/ubench/fft8192_lab 1024 2048 4096 > $OUT/lab.txt 2>&1
tail -60 $OUT/lab.txt
timeout 300 python bench.py --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json
timeout 200 python bench.py --steps 200 --warmup 20 --sets 1 --no-cpu-baseline > $OUT/bench_sets1.json 2> $OUT/bench_sets1.err; tail -c 1500 $OUT/bench_sets1.json
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_20steps.json 2>&1; tail -c 1500 $OUT/bench_20steps.json
timeout 300 python bench.py --gpus 2 --steps 50 --warmup 5 > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; tail -c 1500 $OUT/bench_gpus2.json; tail -5 $OUT/bench_gpus2.err
timeout 300 python bench.py --gpus 2 --shard frames --steps 50 --warmup 5 > $OUT/bench_gpus2_frames.json 2> $OUT/bench_gpus2_frames.err; tail -c 1500 $OUT/bench_gpus2_frames.json; tail -5 $OUT/bench_gpus2_frames.err
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; tail -c 1500 $OUT/bench_cfg3.json; tail -3 $OUT/bench_cfg3.err
timeout 300 python bench.py --config 5 --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 1500 $OUT/bench_cfg5.json; tail -3 $OUT/bench_cfg5.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs head -8
