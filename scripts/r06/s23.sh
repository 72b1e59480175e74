#!/bin/bash
# round 6, session 23 (session 20 once more, on another box): the tail event alone (SS_DRAIN_WAITER_US=0) against the form until now, eight alternations of the 20-step form
OUT=gpurun_out/r06_s23
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="--gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib"
for i in 1 2 3 4 5 6 7 8; do
  for v in event old; do
    E="SS_DRAIN_WAITER_US=0"; [ $v = old ] && E="SS_DRAIN_WAITER_US=0 SS_DRAIN_TAIL_EVENT=0"
    env $E timeout 300 python bench.py $B --steps 20 --warmup 5 > $OUT/k20_${v}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob, statistics
acc = {}
for f in sorted(glob.glob('gpurun_out/r06_s23/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    acc.setdefault(f.split('/')[-1].split('_')[1], []).append(j['ms_per_step'] * 1e3)
for k, v in acc.items():
    print(k, sorted(v), 'median', statistics.median(v))
PY
