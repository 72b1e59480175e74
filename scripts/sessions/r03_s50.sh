#!/bin/bash
# round 3, GPU session 50: what the column tiles of the long transforms pay for their table loads (window, step-A twiddle): A/B builds, garbage results
OUT=gpurun_out/r03_s50; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
for v in base colsnotw colsnowin colsnone; do
  SS_ABLATE_ROLES=3 $B --config 3 --steps 200 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg3_$v.json 2> $OUT/err
  SS_ABLATE_ROLES=3 $B --config 5 --steps 100 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg5_$v.json 2> $OUT/err
  SS_ABLATE_ROLES=3 $B --config 3 --steps 200 --fmt cf32 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg3cf32_$v.json 2> $OUT/err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s50/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
