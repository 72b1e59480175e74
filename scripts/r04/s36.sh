#!/bin/bash
# round 4, session 36: 65536 points with ONE launch per call as the product's default (detect-mode calls of up to 128 frames; column tiles
# first, the row tiles of the call before behind them): the whole GPU suite, the 60-session soak, config 3 by call size
OUT=gpurun_out/r04_s36
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
SS_FUZZ_CULL_SEEDS=60 timeout 1200 python -m pytest tests/test_gpu_cull.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k random > $OUT/pytest_soak.txt 2>&1
echo "soak rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 3"
for f in 16 64 128 256; do
  timeout 300 $B --frames $f --steps $((12800 / f)) > $OUT/c3_f$f.json 2>> $OUT/ab.err
done
timeout 300 $B --frames 128 --steps 100 > $OUT/c3_f128_b.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s36/c3_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        nb = j['config']['frames_per_batch']
        print(os.path.basename(f), nb, j['ms_per_step'], 'us/frame %.3f' % (j['ms_per_step'] * 1e3 / nb), j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -4 $OUT/pytest_gpu.txt | cut -c1-400; tail -2 $OUT/pytest_soak.txt | cut -c1-200; tail -3 $OUT/ab.err | cut -c1-300
