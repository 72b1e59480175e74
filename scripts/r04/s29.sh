#!/bin/bash
# round 4, session 29: 65536 points — calls of more than 256 frames go through in chunks of 256; the reference tests at chunked call
# sizes (65536 x 300, 2^20 x 40); microseconds per frame by call size against SS_CHUNK_65536=0
OUT=gpurun_out/r04_s29
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py -m gpu -q -x -s --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 3 --diag-lib"
for f in 256 384 512 1024; do
  timeout 300 $B --frames $f --steps $((12800 / f)) > $OUT/c3_f${f}_chunk.json 2>> $OUT/ab.err
  SS_CHUNK_65536=0 timeout 300 $B --frames $f --steps $((12800 / f)) > $OUT/c3_f${f}_whole.json 2>> $OUT/ab.err
done
SS_CHUNK_65536=192 timeout 300 $B --frames 384 --steps 33 > $OUT/c3_f384_chunk192.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s29/c3_f*.json'), key=lambda p: (int(p.split('_f')[-1].split('_')[0]), p)):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        nb = j['config']['frames_per_batch']
        print(os.path.basename(f), nb, j['ms_per_step'], 'us/frame %.3f' % (j['ms_per_step'] * 1e3 / nb), j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; grep -h "^\[" $OUT/pytest_gpu.txt | cut -c1-250; tail -3 $OUT/pytest_gpu.txt | cut -c1-300; tail -3 $OUT/ab.err | cut -c1-300
