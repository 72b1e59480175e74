#!/bin/bash
# round 4, session 27: 2^20 points — calls of more than 16 frames go through in chunks of 16 (the chunk's work buffer stays in the Infinity
# Cache between its column and its row half): microseconds per frame by call size against SS_CHUNK_LONG=0; the 2^20-point tests
OUT=gpurun_out/r04_s27
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 5 --diag-lib"
for f in 16 24 32 48 64; do
  timeout 300 $B --frames $f --steps $((1600 / f)) > $OUT/c5_f${f}_chunk.json 2>> $OUT/ab.err
  SS_CHUNK_LONG=0 timeout 300 $B --frames $f --steps $((1600 / f)) > $OUT/c5_f${f}_whole.json 2>> $OUT/ab.err
done
SS_CHUNK_LONG=8 timeout 300 $B --frames 64 --steps 25 > $OUT/c5_f64_chunk8.json 2>> $OUT/ab.err
SS_CHUNK_LONG=12 timeout 300 $B --frames 48 --steps 33 > $OUT/c5_f48_chunk12.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s27/c5_f*.json'), key=lambda p: (int(p.split('_f')[-1].split('_')[0]), p)):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        nb = j['config']['frames_per_batch']
        print(os.path.basename(f), nb, j['ms_per_step'], 'us/frame %.2f' % (j['ms_per_step'] * 1e3 / nb), j['value'], [(k['slot'], k['us'], k['launches_per_call']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -4 $OUT/pytest_gpu.txt | cut -c1-400; tail -3 $OUT/ab.err | cut -c1-300
