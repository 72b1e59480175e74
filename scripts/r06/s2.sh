#!/bin/bash
# round 6, session 2: the fold as a radix-8 butterfly per point (fft65536_dif8.h: dif8_front2_bfly, KIND 8) — lab against round 5's
# accumulating loop, both checked against an fp64 FFT; the config-3 tests on the product library; same-session A/B of config 3 in the
# product (scripts/ab/libspecscan_dif8acc.so = SS_DIF8_BFLY=0); the retune test; where the end of the 20-step form goes (engine sync first?)
OUT=gpurun_out/r06_s2
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
DIF8_ONLY=4 timeout 300 scripts/ubench/dif8_lab 128 256 512 64 > $OUT/dif8_lab_acc.txt 2>&1; cat $OUT/dif8_lab_acc.txt
DIF8_ONLY=6 timeout 300 scripts/ubench/dif8_lab 128 256 512 64 > $OUT/dif8_lab_bfly.txt 2>&1; cat $OUT/dif8_lab_bfly.txt
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -x -q -m gpu -k "config3 or retune or getfft" > $OUT/pytest_cfg3.txt 2>&1; tail -3 $OUT/pytest_cfg3.txt
for i in 1 2; do
  for v in new acc; do
    L=""; [ $v = acc ] && L="--lib scripts/ab/libspecscan_dif8acc.so"; [ $v = new ] && L="--diag-lib"
    for fr in 128 512; do
      timeout 300 python bench.py --config 3 --gpus 1 --frames $fr --steps 100 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub $L > $OUT/cfg3_${v}_f${fr}_$i.json 2>/dev/null
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s2/cfg3_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [k['us'] for k in j['roofline']['kernels']])
    except Exception as e:
        print(f, 'ERR', e)
PY
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k20_dev_$i.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --sync-engine-first > $OUT/k20_eng_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s2/k20_*.json')):
    j = json.loads(open('bench_full.json').read()) if False else json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'])
PY
