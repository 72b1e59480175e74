#!/bin/bash
# round 5, session 23: the fold's launches — the plan workgroups placed by the order pattern (behind the fold's workgroups instead of
# ahead of everything), and calls a little short of a whole round of workgroups (112 / 120 frames: 448 / 480 of the 512 slots, the
# passengers in the rest)
OUT=gpurun_out/r05_s23
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
run() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 200 --warmup 5 --no-cpu-baseline --lib scripts/ab/libspecscan_base.so > $OUT/${tag}_f$f.json 2> $OUT/${tag}_f$f.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_f$f.json').read().strip().splitlines()[-1])
    print('f=$f $tag', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('f=$f $tag ERR', e, open('$OUT/${tag}_f$f.err').read()[-600:])
PY
}
for rep in 1 2; do
  for f in 128 512; do
    run FED_$rep $f SS_X=0
    run FPED_$rep $f 'SS_STEP_ORDER=F*,P*,E*,D*'
    run FEDP_$rep $f 'SS_STEP_ORDER=F*,E*,D*,P*'
  done
  for f in 96 104 112 120 124 128 240 248 256; do
    run FED_$rep $f SS_X=0
  done
done
