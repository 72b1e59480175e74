#!/bin/bash
# round 5, session 28: the fold's rows in blocks of 32 Q bins (a tile column's 256 bins one run of 1 KB instead of eight lines 32 KB apart
# on one memory channel) — parity of the 65536 / 131072-point forms, then the orders of session 25 / 26 again, stamps
OUT=gpurun_out/r05_s28
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
S=$SECONDS
timeout 900 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py -x -q -m gpu -k "shipped_form or getfft or cull" > $OUT/pytest.txt 2>&1
echo "pytest rc=$? $((SECONDS-S)) s"; tail -5 $OUT/pytest.txt
LIB=scripts/ab/libspecscan_base.so
run() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 200 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/${tag}_f$f.json 2> $OUT/${tag}_f$f.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_f$f.json').read().strip().splitlines()[-1])
    print('f=$f $tag', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('f=$f $tag ERR', e, open('$OUT/${tag}_f$f.err').read()[-600:])
PY
}
for f in 128 64 256 512 16 32; do
  run FED $f SS_X=0
  run FPED $f 'SS_STEP_ORDER=F*,P*,E*,D*'
  run D64FPED $f 'SS_STEP_ORDER=D64,F*,P*,E*,D*'
  run FPED_l0 $f 'SS_STEP_ORDER=F*,P*,E*,D*' SS_LIST_FIRST=0
done
stamps() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" SS_STEP_STAMPS=$OUT/stamps_${tag}_f$f.txt timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/st_${tag}_f$f.json 2> $OUT/st_${tag}_f$f.err
  echo "== stamps $tag, $f frames"
  python scripts/analyze_step_stamps.py $OUT/stamps_${tag}_f$f.txt 2>&1 | tee $OUT/stamps_${tag}_f${f}_summary.txt
}
stamps FED 128 SS_X=0
stamps D64FPED 128 'SS_STEP_ORDER=D64,F*,P*,E*,D*'
stamps FPED_l0 64 'SS_STEP_ORDER=F*,P*,E*,D*' SS_LIST_FIRST=0
