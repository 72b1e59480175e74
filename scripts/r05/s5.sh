#!/bin/bash
# round 5, session 5: the fold's launch and its passengers — one residue per workgroup (8 waves per SIMD) against two (4 waves), wave
# priorities, where the listed tiles ride, the dispatch order — config 3 at 128- and 512-frame calls
OUT=gpurun_out/r05_s5
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
for lib in dif8w8 dif8w4; do
  for f in 128 512; do
    run base $lib $f SS_X=0
    run prio2 $lib $f SS_STEP_PRIO_OTHER=2
    run list0 $lib $f SS_LIST_FIRST=0
    run orderFD $lib $f 'SS_STEP_ORDER=E*|F*,D*'
    run orderFDprio $lib $f 'SS_STEP_ORDER=E*|F*,D*' SS_STEP_PRIO_OTHER=2
    run prio3fft1 $lib $f SS_STEP_PRIO_OTHER=3 SS_STEP_PRIO_FFT=1
  done
done
