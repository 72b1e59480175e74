#!/bin/bash
# round 4, session 26: 2^20 points — microseconds per frame by call size (does the work buffer's stay in the 256 MiB Infinity Cache pay?)
OUT=gpurun_out/r04_s26
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 5"
for f in 4 8 12 16 24 32 64; do
  timeout 300 $B --frames $f --steps $((1600 / f)) > $OUT/c5_f$f.json 2>> $OUT/ab.err
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s26/c5_f*.json'), key=lambda p: int(p.split('_f')[-1].split('.')[0])):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        nb = j['config']['frames_per_batch']
        print(os.path.basename(f), nb, j['ms_per_step'], 'us/frame %.2f' % (j['ms_per_step'] * 1e3 / nb), j['value'], [(k['slot'], k['us'], round(k['us'] / nb, 2)) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
tail -3 $OUT/ab.err | cut -c1-300
