#!/bin/bash
# round 3, GPU session 45: ring rows at the drain against ring rows by every call's tiles, alternating, 200 and 20 steps
OUT=gpurun_out/r03_s45; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
for v in base ringlegacy; do
timeout 300 python bench.py --no-cpu-baseline --no-also --lib scripts/ab/libspecscan_$v.so > $OUT/${v}_k200_r$rep.json 2> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --lib scripts/ab/libspecscan_$v.so > $OUT/${v}_k20_r$rep.json 2> $OUT/bench.err
done
done
python - <<'PY'
import json, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r03_s45/*.json')):
    j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    acc[os.path.basename(f).rsplit('_r', 1)[0]].append(j['ms_per_step'] * 1e3)
for k, v in sorted(acc.items()):
    print(k, ' '.join(f'{x:.1f}' for x in v), 'mean %.2f' % (sum(v) / len(v)))
PY
