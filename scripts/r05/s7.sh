#!/bin/bash
# round 5, session 7: what each passenger costs the fold's launch (SS_ABLATE_ROLES of the diagnostics build: 1 = no detect role, 2 = no
# emit role; garbage results, timing only), config 3 at 128 / 512 frames, two residues per workgroup, order E*|F*,D*
OUT=gpurun_out/r05_s7
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
for lib in dif8w4 dif8w8; do
  for f in 128 512; do
    run all $lib $f 'SS_STEP_ORDER=E*|F*,D*'
    run nodet $lib $f 'SS_STEP_ORDER=E*|F*,D*' SS_ABLATE_ROLES=1
    run noemit $lib $f 'SS_STEP_ORDER=E*|F*,D*' SS_ABLATE_ROLES=2
    run neither $lib $f 'SS_STEP_ORDER=E*|F*,D*' SS_ABLATE_ROLES=3
    run nocull $lib $f 'SS_STEP_ORDER=E*|F*,D*' SS_CULL=0
  done
done
