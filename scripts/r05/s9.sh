#!/bin/bash
# round 5, session 9: the fold's plan rewritten (plan_dif8_run: 32 columns x one frame tile per block, coalesced) — the config 3 tests
# and the culling tests, then config 3 at 128 / 256 / 512 frames, one and two residues per workgroup, two dispatch orders
OUT=gpurun_out/r05_s9
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py -m gpu -x -q -k "config3 or 65536 or getfft" > $OUT/pytest_65536.txt 2>&1
tail -4 $OUT/pytest_65536.txt | cut -c1-300
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
  for f in 128 256 512 64; do
    run EFD $lib $f SS_X=0
    run FED $lib $f 'SS_STEP_ORDER=F*,E*,D*'
  done
  run EFD_neither $lib 128 SS_ABLATE_ROLES=3
  run EFD_neither $lib 512 SS_ABLATE_ROLES=3
done
