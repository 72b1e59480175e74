#!/bin/bash
# round 2, GPU session 17: the column half of the long transforms as the FFT role of k_scan_step (configs 3 and 5):
# parity first, then the step time with and without the overlap.
set -x
OUT=gpurun_out/r02_s17; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_step_pipeline.py tests/test_gpu_stated_configs.py tests/test_gpu_parity.py -x -q -m gpu -s > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
for cfg in 3 5; do
  timeout 300 python bench.py --config $cfg --gpus 1 --no-cpu-baseline > $OUT/bench_cfg$cfg.json 2> $OUT/bench_cfg$cfg.err; cat $OUT/bench_cfg$cfg.json
  SS_PIPELINE=0 timeout 300 python bench.py --diag-lib --config $cfg --gpus 1 --no-cpu-baseline > $OUT/bench_cfg${cfg}_nopipe.json 2> $OUT/bench_cfg${cfg}_nopipe.err; cat $OUT/bench_cfg${cfg}_nopipe.json
done
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json
