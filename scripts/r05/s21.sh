#!/bin/bash
# round 5, session 21: the driver's 20-step form of the default line — launch queues of the deep pipeline (2 ship; 3, 4), and what the one
# timed launch costs it — alternating runs, diagnostics build; 65536-point calls of 16 / 32 / 64 frames in the fold and in round 4's forms
OUT=gpurun_out/r05_s21
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
run() {  # tag steps extra-args env...
  tag=$1; k=$2; extra=$3; shift 3
  env "$@" timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib $extra > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1])
    print('$tag', j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('kernel_us'), j['roofline'].get('kernel_us_behind_the_timed_region'))
except Exception as e:
    print('$tag ERR', e, open('$OUT/$tag.err').read()[-400:])
PY
}
for rep in 1 2 3; do
  run q2_k20_$rep 20 "" SS_X=0
  run q3_k20_$rep 20 "--sets 12" SS_QUEUES=3
  run q2_k20_notiming_$rep 20 "--no-kernel-timing" SS_X=0
  run q2_k200_$rep 200 "" SS_X=0
  run q3_k200_$rep 200 "--sets 12" SS_QUEUES=3
done
for f in 16 32 64; do
  for mode in fold four; do
    if [ $mode = four ]; then e="SS_DIF8=0"; else e="SS_X=0"; fi
    env $e timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 200 --warmup 5 --no-cpu-baseline --diag-lib > $OUT/c3_${mode}_f$f.json 2> $OUT/c3_${mode}_f$f.err
    python - <<PY
import json
try:
    j = json.loads(open('$OUT/c3_${mode}_f$f.json').read().strip().splitlines()[-1])
    print('cfg 3 f=$f $mode', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('cfg 3 f=$f $mode ERR', e, open('$OUT/c3_${mode}_f$f.err').read()[-400:])
PY
  done
done
