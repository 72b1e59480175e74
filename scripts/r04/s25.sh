#!/bin/bash
# round 4, session 25: the ring placement as a function of its own (csrc/ring_place.h: protected spans instead of one comparison) —
# the whole GPU suite, the 60-session soak, the ring-wrap cases with detect(k - 1) riding on the row launch (SS_DET_LAG2=0: the form in
# which the old comparison let a batch land on the window of the call before), configs 3 and 5 once more
OUT=gpurun_out/r04_s25
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
SS_FUZZ_CULL_SEEDS=60 timeout 1200 python -m pytest tests/test_gpu_cull.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k random > $OUT/pytest_soak.txt 2>&1
echo "soak rc=$?" >> $OUT/rc.txt
SS_DET_LAG2=0 timeout 1200 python -m pytest tests/test_gpu_cull.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "device_calls or long_rows" > $OUT/pytest_lag1.txt 2>&1
echo "lag1 rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --config 5 --steps 100 > $OUT/c5_$rep.json 2>> $OUT/ab.err
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s25/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -4 $OUT/pytest_gpu.txt | cut -c1-400; tail -3 $OUT/pytest_soak.txt | cut -c1-300; tail -3 $OUT/pytest_lag1.txt | cut -c1-300
