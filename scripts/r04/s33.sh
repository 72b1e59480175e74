#!/bin/bash
# round 4, session 33: 65536 x 128 — every other column tile (group of row tiles) starts a few microseconds late: do the load, compute
# and store phases of a launch that is ONE round of workgroups overlap better out of step? (A/B builds, scripts/build_ab.py)
OUT=gpurun_out/r04_s33
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 3 --steps 100"
for rep in 1 2; do
  for v in base colsstag1 colsstag2 colsstag3 rowsstag1 rowsstag2 bothstag2; do
    timeout 300 $B --lib scripts/ab/libspecscan_$v.so > $OUT/c3_${v}_$rep.json 2>> $OUT/ab.err
  done
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s33/c3_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
tail -3 $OUT/ab.err | cut -c1-300
