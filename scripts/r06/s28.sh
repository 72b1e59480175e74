#!/bin/bash
# round 6, session 28: does a longer untimed preheat change the 20-step form on this box (clocks)? 400 ms (default) / 1500 / 4000, alternating
OUT=gpurun_out/r06_s28
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="--gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --steps 20 --warmup 5"
for i in 1 2 3 4 5 6; do
  for p in 400 1500 4000; do
    timeout 300 python bench.py $B --preheat-ms $p > $OUT/k20_p${p}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob, statistics
acc = {}
for f in sorted(glob.glob('gpurun_out/r06_s28/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    acc.setdefault(f.split('/')[-1].split('_')[1], []).append(round(j['ms_per_step'] * 1e3, 2))
for k, v in acc.items():
    print(k, sorted(v), 'median', statistics.median(v))
PY
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | head -30
