#!/bin/bash
# round 4, session 21: 65536 points — where the emit stage rides (column launch / row launch) and how wide it is (one workgroup per
# frame / one wave per frame); the new ring-wrap cases of test_gpu_cull.py
OUT=gpurun_out/r04_s21
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cull.py tests/test_gpu_stated_configs.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_new_$rep.json 2>> $OUT/ab.err
  SS_EMIT_ON_ROWS=1 timeout 300 $B --config 3 --steps 100 > $OUT/c3_emitrows_$rep.json 2>> $OUT/ab.err
  SS_EMIT_WIDE=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_narrow_$rep.json 2>> $OUT/ab.err
  SS_LIST_FIRST=32 timeout 300 $B --config 3 --steps 100 > $OUT/c3_first32_$rep.json 2>> $OUT/ab.err
done
for v in new emitrows narrow; do
  E=""; [ $v = emitrows ] && E="SS_EMIT_ON_ROWS=1"; [ $v = narrow ] && E="SS_EMIT_WIDE=0"
  env $E timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_$v.json 2>> $OUT/ab.err
  env $E timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_$v.json 2>> $OUT/ab.err
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s21/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -8 $OUT/pytest_gpu.txt | cut -c1-600; tail -3 $OUT/ab.err | cut -c1-300
