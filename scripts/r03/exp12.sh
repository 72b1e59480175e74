#!/bin/bash
OUT=gpurun_out/r03_s18
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --warmup 5 --diag-lib --steps 200 --start-level 100"
timeout 300 $B > $OUT/bench_sl100.json 2> $OUT/bench_sl100.err
SS_PLAN_NOZERO=1 timeout 300 $B > $OUT/bench_sl100_nozero.json 2> $OUT/bench_sl100_nozero.err
SS_PLAN_NOZERO=1 SS_HINT_MODE=1 timeout 300 $B > $OUT/bench_sl100_nozero_hint1.json 2> $OUT/bench_sl100_nozero_hint1.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s18/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
