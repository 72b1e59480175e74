#!/bin/bash
# round 5, session 25: the fold's launches after session 24's stamps — the plan's loads batched; the plan workgroups behind the fold's
# (order F*,P*,E*,D*); the listed tiles behind the fold workgroups' own transforms (SS_LIST_FIRST=0) instead of on detect workgroups that
# wait for a slot; stamps of the candidates at 128 frames, and of the passengers by themselves (SS_HINT_MODE=4)
OUT=gpurun_out/r05_s25
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
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
for f in 128 64 256 512 16; do
  run FED $f SS_X=0
  run FPED $f 'SS_STEP_ORDER=F*,P*,E*,D*'
  run FED_l0 $f SS_LIST_FIRST=0
  run FPED_l0 $f 'SS_STEP_ORDER=F*,P*,E*,D*' SS_LIST_FIRST=0
  run FEPD_l0 $f 'SS_STEP_ORDER=F*,E*,P*,D*' SS_LIST_FIRST=0
done
stamps() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" SS_STEP_STAMPS=$OUT/stamps_${tag}_f$f.txt timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/st_${tag}_f$f.json 2> $OUT/st_${tag}_f$f.err
  echo "== stamps $tag, $f frames"
  python scripts/analyze_step_stamps.py $OUT/stamps_${tag}_f$f.txt 2>&1 | tee $OUT/stamps_${tag}_f${f}_summary.txt
}
stamps FED 128 SS_X=0
stamps FPED_l0 128 'SS_STEP_ORDER=F*,P*,E*,D*' SS_LIST_FIRST=0
stamps alone 128 SS_HINT_MODE=4
stamps FPED_l0 64 'SS_STEP_ORDER=F*,P*,E*,D*' SS_LIST_FIRST=0
