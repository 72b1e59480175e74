#!/bin/bash
# round 2, GPU session 23: deep pipelining (launch L = FFT(L) + detect(L-2) + emit(L-4), alternating over two queues)
set -x
OUT=gpurun_out/r02_s23; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_step_pipeline.py tests/test_gpu_parity.py tests/test_gpu_stated_configs.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --diag-lib --no-cpu-baseline 2> $OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'], d['roofline']['kernel_us'], d.get('parity'))" | tee -a $OUT/summary.txt
}
for rep in 1 2; do
run deep SS_X=0
run roles_in_order SS_DEEP=0
done
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cat $OUT/bench_k20.json
