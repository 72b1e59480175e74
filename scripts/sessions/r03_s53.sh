#!/bin/bash
# round 3, GPU session 53 (last): 65536 points culled against unculled on one box, alternating, 128- and 16-frame calls
OUT=gpurun_out/r03_s53; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="timeout 100 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 100 --sub --config 3"
for rep in 1 2 3; do
  $B --steps 200 > $OUT/f128_cull_r$rep.json 2> $OUT/err
  $B --steps 200 --no-cull > $OUT/f128_nocull_r$rep.json 2> $OUT/err
done
$B --steps 300 --frames 16 > $OUT/f16_cull_r1.json 2> $OUT/err
$B --steps 300 --frames 16 --no-cull > $OUT/f16_nocull_r1.json 2> $OUT/err
$B --steps 300 --frames 32 > $OUT/f32_cull_r1.json 2> $OUT/err
$B --steps 300 --frames 32 --no-cull > $OUT/f32_nocull_r1.json 2> $OUT/err
python - <<'PY'
import json, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r03_s53/*.json')):
    j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    acc[os.path.basename(f).rsplit('_r', 1)[0]].append((j['ms_per_step'] * 1e3, j['value'] / 1e3, {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}))
for k, v in sorted(acc.items()):
    print(k, ' '.join(f'{x[0]:.1f}' for x in v), 'us per call;', ' '.join(f'{x[1]:.1f}' for x in v), 'GS/s;', v[0][2])
PY
