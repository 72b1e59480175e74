#!/bin/bash
# round 2, GPU session 18: long transforms — one launch per stage (SS_STEP_LONG=0) against the step kernel, and dispatch orders
set -x
OUT=gpurun_out/r02_s18; mkdir -p $OUT
run() {  # name config env...
  local name=$1 cfg=$2; shift 2
  env "$@" timeout 300 python bench.py --diag-lib --config $cfg --gpus 1 --no-cpu-baseline 2> $OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])" | tee -a $OUT/summary.txt
}
for rep in 1 2; do
for cfg in 3 5; do
  run cfg${cfg}_stages $cfg SS_STEP_LONG=0
  run cfg${cfg}_step_default $cfg SS_X=0
  run cfg${cfg}_step_nopipe $cfg SS_PIPELINE=0
  run cfg${cfg}_order_F_then_D $cfg 'SS_STEP_ORDER=E*|F*,D*'
  run cfg${cfg}_order_D_then_F $cfg 'SS_STEP_ORDER=E*|D*,F*'
  run cfg${cfg}_order_D1F1 $cfg 'SS_STEP_ORDER=E*|D1,F1'
  run cfg${cfg}_order_D2F1 $cfg 'SS_STEP_ORDER=E*|D2,F1'
  run cfg${cfg}_order_D64F64 $cfg 'SS_STEP_ORDER=E*|D64,F64'
  run cfg${cfg}_order_D512F512 $cfg 'SS_STEP_ORDER=E*|D512,F512'
  run cfg${cfg}_order_F512D512 $cfg 'SS_STEP_ORDER=E*|F512,D512'
done
done
