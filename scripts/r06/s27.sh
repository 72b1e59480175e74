#!/bin/bash
# round 6, session 27: the interpreter's collector switched off inside the timed region (as timeit does) against left on (SS_BENCH_KEEP_GC=1):
# ten alternations of the 20-step form
OUT=gpurun_out/r06_s27
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="--gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python bench.py $B > $OUT/k20_nogc_$i.json 2>/dev/null
  SS_BENCH_KEEP_GC=1 timeout 300 python bench.py $B > $OUT/k20_gc_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob, statistics
acc = {}
for f in sorted(glob.glob('gpurun_out/r06_s27/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    acc.setdefault(f.split('/')[-1].split('_')[1], []).append(round(j['ms_per_step'] * 1e3, 2))
for k, v in acc.items():
    print(k, sorted(v), 'median', statistics.median(v))
PY
