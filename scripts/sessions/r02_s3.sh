#!/bin/bash
# round 2, GPU session 3: dispatch-order patterns of k_scan_step, other configs, PMC traffic, lab with a proper working set
set -x
OUT=gpurun_out/r02_s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
run_order() {
  SS_STEP_ORDER="$1" timeout 200 python bench.py --diag-lib --steps 200 --warmup 20 --no-cpu-baseline > $OUT/order.json 2>&1
  python - "$OUT/order.json" "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print("ORDER %-28s %9.1f MS/s %.4f ms/step kernel %.2f us" % (sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['kernel_us']))
except Exception as e: print("ORDER",sys.argv[2],"ERR",e)
PY
}
for o in "E*|D128,F128" "E*|D128,F128" "E*|D128,F256" "F512,E*|D128,F128" "F256,E*|D128,F128" "F384,E*|D128,F128" "E*,F128|D128,F128" "|D128,F128,E16" "|F128,D128,E16" \
         "E*|D96,F96" "E*|D160,F160" "E*|D192,F192" "E*|D128,F64" "E*|D64,F128" "E*|D256,F128" "E*|D384,F128" "F*|D*,E*" "F768|D128,F64,E16" "E*|D128,F192" "E*|D120,F120" "E*|D136,F136" \
         "D128,E*|F128,D128" "E*|D32,F32" "E*|D16,F16" "E*|D2,F2" "F640|D128,F128,E128"; do run_order "$o"; done 2>&1 | grep ORDER | tee $OUT/orders.txt
timeout 300 scripts/ubench/fft8192_lab 1024 > $OUT/lab1024.txt 2>&1; tail -16 $OUT/lab1024.txt
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; tail -c 900 $OUT/bench_cfg3.json
timeout 300 python bench.py --config 5 --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; tail -c 900 $OUT/bench_cfg5.json
timeout 300 python bench.py --fft 4096 --frames 2048 --sample-rate 1024000 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_4096.json 2>&1; tail -c 600 $OUT/bench_4096.json
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT; ls $OUT/pmc_fetch/*/ | head
