#!/bin/bash
# round 6, session 13: 65536 points, short calls — one residue per workgroup (KIND 11, calls of <= 32 frames) against two (SS_DIF8_SINGLE_MAX=0),
# alternating; the 65536-point tests (cull, stated configs, fuzz) on the new binaries
OUT=gpurun_out/r06_s13
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest_long.txt 2>&1; tail -3 $OUT/pytest_long.txt
run() { timeout 300 python bench.py --gpus 1 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib ${@:2} > $OUT/$1.json 2>/dev/null; }
for i in 1 2; do
  for fr in 8 16 32 64; do
    run f${fr}_single_$i --config 3 --frames $fr --steps 150
    SS_DIF8_SINGLE_MAX=0 run f${fr}_two_$i --config 3 --frames $fr --steps 150
    SS_DIF8_SINGLE_MAX=64 run f${fr}_single64_$i --config 3 --frames $fr --steps 150
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s13/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(f, 'ERR', e)
PY
