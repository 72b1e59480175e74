#!/bin/bash
# round 3, GPU session 48: the end of the timed region — the contract's device-wide synchronisation as the only wait, against ss_sync first
OUT=gpurun_out/r03_s48; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > $OUT/new_k20_r$rep.json 2> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --sync-engine-first > $OUT/old_k20_r$rep.json 2> $OUT/bench.err
done
timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/new_k200.json 2> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-also --sync-engine-first > $OUT/old_k200.json 2> $OUT/bench.err
python - <<'PY'
import json, glob, os, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r03_s48/*.json')):
    j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    acc[os.path.basename(f).rsplit('_r', 1)[0].replace('.json', '')].append((j['ms_per_step'] * 1e3, j['config']['tail_us']))
for k, v in sorted(acc.items()):
    print(k, ' '.join(f'{x[0]:.1f}' for x in v), 'mean %.2f' % (sum(x[0] for x in v) / len(v)), v[0][1])
PY
