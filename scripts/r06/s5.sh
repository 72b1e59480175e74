#!/bin/bash
# round 6, session 5: (a) the long transforms' big stores write-through (work buffer, dB / ring rows: SS_AUX_WORK / SS_AUX_ROWS = 16)
# against the default policy (scripts/ab/libspecscan_longwb.so) — 262144 x 32 / x 64, 2^20 x 16, 65536 x 128 CF32 — alternating runs;
# (b) the radix-16 fold as a butterfly per point (131072 points): lab against the accumulating form, tests, rate; (c) the long-transform
# tests on the new binaries
OUT=gpurun_out/r06_s5
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
DIF_Q=16 DIF8_ONLY=5 timeout 300 scripts/ubench/dif8_lab 64 128 256 > $OUT/dif16_lab_acc.txt 2>&1; cat $OUT/dif16_lab_acc.txt
DIF_Q=16 DIF8_ONLY=7 timeout 300 scripts/ubench/dif8_lab 64 128 256 > $OUT/dif16_lab_bfly.txt 2>&1; cat $OUT/dif16_lab_bfly.txt
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py -x -q -m gpu > $OUT/pytest_long.txt 2>&1; tail -3 $OUT/pytest_long.txt
run() { # name, lib flag, bench args
  timeout 300 python bench.py --gpus 1 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub $2 ${@:3} > $OUT/$1.json 2>/dev/null
}
for i in 1 2; do
  for v in wt wb; do
    L="--diag-lib"; [ $v = wb ] && L="--lib scripts/ab/libspecscan_longwb.so"
    run x256_f32_${v}_$i "$L" --config 5 --fft 262144 --frames 32 --steps 60
    run x256_f64_${v}_$i "$L" --config 5 --fft 262144 --frames 64 --steps 40
    run cfg5_${v}_$i "$L" --config 5 --steps 60
    run cf32_65536_${v}_$i "$L" --config 3 --fmt cf32 --steps 100
    run cfg3_${v}_$i "$L" --config 3 --steps 100
    run n131072_${v}_$i "$L" --config 3 --fft 131072 --frames 64 --steps 60
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s5/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config']['tile_culling'])
    except Exception as e:
        print(f, 'ERR', e)
PY
