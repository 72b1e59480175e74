#!/bin/bash
# round 3, GPU session 49: the benchmark line of the final bench.py (the driver's forms), beside s47's evidence of the same library
OUT=gpurun_out/r03_s49; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-220 $OUT/bench_default.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cut -c1-220 $OUT/bench_k20.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
