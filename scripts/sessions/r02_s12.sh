#!/bin/bash
# round 2, GPU session 12: long rows — XCD-aware detect tile order, several waves per frame in the emit stage, larger ring
set -x
OUT=gpurun_out/r02_s12; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v
SS_DET_XCD_ORDER=$1 SS_EMIT_WIDE=$2 timeout 300 python bench.py --diag-lib --config 3 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/cfg3_$1_$2.json 2>&1
SS_DET_XCD_ORDER=$1 SS_EMIT_WIDE=$2 timeout 300 python bench.py --diag-lib --config 5 --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline > $OUT/cfg5_$1_$2.json 2>&1
SS_DET_XCD_ORDER=$1 SS_EMIT_WIDE=$2 timeout 300 python bench.py --diag-lib --fft 65536 --frames 128 --sample-rate 20000000 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/n65536cf32_$1_$2.json 2>&1
SS_DET_XCD_ORDER=$1 SS_EMIT_WIDE=$2 timeout 300 python bench.py --diag-lib --fft 16384 --frames 512 --sample-rate 4096000 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/n16384_$1_$2.json 2>&1
done
for f in $OUT/*_?_?.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print("%-34s %9.1f MS/s %.4f ms/step" % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step']))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_cfg3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_cfg3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_cfg5.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch_cfg3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 20 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_fetch_cfg3.log 2>&1
cd $GRAFT_REPO_ROOT; find $OUT -name "*kernel_stats.csv" | xargs -n1 head -6 | cut -c1-160
