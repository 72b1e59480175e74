#!/bin/bash
# round 5, session 6: dispatch orders of the fold's launch (plan workgroups always first): passengers spread between the fold
# workgroups instead of in front of / behind them — config 3 at 128-, 256- and 512-frame calls, one and two residues per workgroup
OUT=gpurun_out/r05_s6
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
    print('$lib f=$f $tag', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('$lib f=$f $tag ERR', e, open('$OUT/${tag}_${lib}_f$f.err').read()[-600:])
PY
}
for lib in dif8w4 dif8w8; do
  for f in 128 256 512; do
    run EFD $lib $f 'SS_STEP_ORDER=E*|F*,D*'
    run F8E2D1 $lib $f 'SS_STEP_ORDER=|F8,E2,D1'
    run F16E4D1 $lib $f 'SS_STEP_ORDER=|F16,E4,D1'
    run DF4E1 $lib $f 'SS_STEP_ORDER=D*|F4,E1'
    run FED $lib $f 'SS_STEP_ORDER=F*,E*,D*'
    run F4E1D $lib $f 'SS_STEP_ORDER=|F4,E1'
  done
done
