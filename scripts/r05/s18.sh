#!/bin/bash
# round 5, session 18: what the noise-ceiling loads of the row-writing kernels cost (A/B builds with the subtraction taken out: garbage
# results, timing only) — config 5 (2^20 x 16) and config 3 (the fold) at 128 / 512 frames, alternating runs
OUT=gpurun_out/r05_s18
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
run() {  # tag lib config frames
  timeout 300 python bench.py --config $3 --frames $4 --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib scripts/ab/libspecscan_$2.so > $OUT/$1_$2_c$3_f$4.json 2> $OUT/$1_$2_c$3_f$4.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/$1_$2_c$3_f$4.json').read().strip().splitlines()[-1])
    print('$2 cfg $3 f=$4 $1', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('$2 cfg $3 f=$4 ERR', e, open('$OUT/$1_$2_c$3_f$4.err').read()[-400:])
PY
}
for rep in 1 2; do
  run r$rep base 5 16
  run r$rep rows1024nothr 5 16
  run r$rep base 3 128
  run r$rep foldnothr 3 128
  run r$rep base 3 512
  run r$rep foldnothr 3 512
done
