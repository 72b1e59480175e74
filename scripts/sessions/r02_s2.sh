#!/bin/bash
# round 2, GPU session 2: k_scan_step (stage pipelining) — GPU test suite, bench, dispatch-order sweep, rocprof
set -x
OUT=gpurun_out/r02_s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
timeout 300 python bench.py --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 2500 $OUT/bench_default.json
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sync-every-step > $OUT/bench_sync_every.json 2>&1; tail -c 600 $OUT/bench_sync_every.json
timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-psd-out > $OUT/bench_detect_mode.json 2>&1; tail -c 600 $OUT/bench_detect_mode.json
for combo in "0 0" "256 256" "128 128" "64 64" "256 512" "512 256" "32 32" "8 8" "1 1" "512 512" "1024 1024" "128 256"; do
  set -- $combo
  SS_STEP_RUN_DET=$1 SS_STEP_RUN_FFT=$2 timeout 200 python bench.py --diag-lib --steps 200 --warmup 20 --no-cpu-baseline > $OUT/order_$1_$2.json 2>&1
  python - "$OUT/order_$1_$2.json" "$1 $2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print("ORDER",sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['kernel_us'])
except Exception as e: print("ORDER",sys.argv[2],"ERR",e)
PY
done
SS_PIPELINE=0 timeout 200 python bench.py --diag-lib --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_nopipeline.json 2>&1; tail -c 600 $OUT/bench_nopipeline.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs head -8
