#!/bin/bash
# round 4, session 28: 65536 points — microseconds per frame by call size (work buffer: 64 MiB per 128 frames)
OUT=gpurun_out/r04_s28
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 3"
for f in 64 128 192 256 384 512; do
  timeout 300 $B --frames $f --steps $((12800 / f)) > $OUT/c3_f$f.json 2>> $OUT/ab.err
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s28/c3_f*.json'), key=lambda p: int(p.split('_f')[-1].split('.')[0])):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        nb = j['config']['frames_per_batch']
        print(os.path.basename(f), nb, j['ms_per_step'], 'us/frame %.3f' % (j['ms_per_step'] * 1e3 / nb), j['value'], [(k['slot'], k['us'], round(k['us'] / nb, 3)) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
tail -3 $OUT/ab.err | cut -c1-300
