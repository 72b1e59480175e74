#!/bin/bash
# round 4, session 20: 65536 points — detect(k - 2) on the COLUMN launch of call k (det_lag2; the plan it needs rode on the column launch
# of call k - 1), the row launch carrying nothing, against session 19's form (SS_DET_LAG2=0: detect(k - 1) on the row launch) and the
# unculled form; the whole GPU suite on this tree
OUT=gpurun_out/r04_s20
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_new_$rep.json 2>> $OUT/ab.err
  SS_DET_LAG2=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_lag1_$rep.json 2>> $OUT/ab.err
  SS_CULL_65536=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_nocull_$rep.json 2>> $OUT/ab.err
done
timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_new.json 2>> $OUT/ab.err
SS_CULL_65536=0 timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_nocull.json 2>> $OUT/ab.err
timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_new.json 2>> $OUT/ab.err
SS_CULL_65536=0 timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_nocull.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s20/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -8 $OUT/pytest_gpu.txt | cut -c1-600; tail -3 $OUT/ab.err | cut -c1-300
