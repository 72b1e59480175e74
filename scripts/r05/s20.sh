#!/bin/bash
# round 5, session 20: ring rows as dB values against the tree before (scripts/ab/libspecscan_base.so, built in session 18: the FFT stage
# subtracts the ceiling) — same session, alternating runs, diagnostics builds both
OUT=gpurun_out/r05_s20
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
run() {  # tag libarg config frames extra
  tag=$1; lib=$2; c=$3; f=$4; shift 4
  timeout 300 python bench.py --config $c --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline $lib "$@" > $OUT/${tag}_c${c}_f$f.json 2> $OUT/${tag}_c${c}_f$f.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_c${c}_f$f.json').read().strip().splitlines()[-1])
    print('$tag cfg $c f=$f $*', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('$tag cfg $c f=$f ERR', e, open('$OUT/${tag}_c${c}_f$f.err').read()[-500:])
PY
}
for rep in 1 2; do
  for cf in "3 128" "3 512" "5 16"; do
    run before$rep "--lib scripts/ab/libspecscan_base.so" $cf
    run dbrows$rep "--diag-lib" $cf
  done
done
