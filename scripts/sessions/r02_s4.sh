#!/bin/bash
# round 2, GPU session 4: order in kernel arguments vs table, repeated; new parity cases on the stated configs
set -x
OUT=gpurun_out/r02_s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -m gpu -x -q -s > $OUT/pytest_stated.txt 2>&1; tail -25 $OUT/pytest_stated.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
run_order() {
  SS_STEP_ORDER="$1" SS_STEP_ORDER_TABLE="$2" timeout 200 python bench.py --diag-lib --steps 300 --warmup 20 --no-cpu-baseline > $OUT/order.json 2>&1
  python - "$OUT/order.json" "$1 table=$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print("ORDER %-36s %9.1f MS/s %.4f ms/step kernel %.2f us" % (sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['kernel_us']))
except Exception as e: print("ORDER",sys.argv[2],"ERR",e)
PY
}
for rep in 1 2; do
for o in "E*|D128,F128" "E*|D256,F256" "E*|D128,F256" "F*|D*,E*" "E*,D*|F*" "E*|D128,F512" "E*,D128,F256|D128,F128" "E*,D128|F256,D128" "D128,E*|F128,D128" "E*|D64,F64" "E*|D192,F192" "E*,D384|F128,D128" "E64,D192|F128,D128,E8"; do run_order "$o" 0; done
run_order "E*|D128,F128" 1
done 2>&1 | grep ORDER | tee $OUT/orders.txt
timeout 300 python bench.py --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
