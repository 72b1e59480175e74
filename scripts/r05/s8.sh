#!/bin/bash
# round 5, session 8: the plan as a launch of its own (SS_PLAN_FUSED=0) against the plan as the first workgroups of the fold's launch
OUT=gpurun_out/r05_s8
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
run() {  # tag lib frames env...
  tag=$1; lib=$2; f=$3; shift 3
  env "$@" timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib scripts/ab/libspecscan_$lib.so > $OUT/${tag}_${lib}_f$f.json 2> $OUT/${tag}_${lib}_f$f.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_${lib}_f$f.json').read().strip().splitlines()[-1])
    print('$lib f=$f $tag', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config']['tiles']['evaluated_frac'])
except Exception as e:
    print('$lib f=$f $tag ERR', e, open('$OUT/${tag}_${lib}_f$f.err').read()[-600:])
PY
}
for lib in dif8w4; do
  for f in 128 512; do
    run fused $lib $f 'SS_STEP_ORDER=F*,E*,D*'
    run planown $lib $f 'SS_STEP_ORDER=F*,E*,D*' SS_PLAN_FUSED=0
    run planown_neither $lib $f 'SS_STEP_ORDER=F*,E*,D*' SS_PLAN_FUSED=0 SS_ABLATE_ROLES=3
  done
done
