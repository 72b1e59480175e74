#!/bin/bash
# round 2, GPU session 25: deep pipelining with the halo frames transformed again — whole GPU suite, smoke, bench
set -x
OUT=gpurun_out/r02_s25; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cat $OUT/bench_k20.json
