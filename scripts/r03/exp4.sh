#!/bin/bash
OUT=gpurun_out/r03_s4
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --steps 200 --warmup 5 --diag-lib"
timeout 300 $B > $OUT/bench_diag.json 2> $OUT/bench_diag.err
for m in 1 2 3; do
SS_ABLATE_ROLES=$m timeout 300 $B > $OUT/bench_ablate$m.json 2> $OUT/bench_ablate$m.err
done
SS_ABLATE_ROLES=3 SS_CULL=0 timeout 300 $B > $OUT/bench_ablate3_nocull.json 2> $OUT/bench_ablate3_nocull.err
SS_ABLATE_ROLES=3 SS_STEP_STAMPS=$OUT/stamps3.txt timeout 300 $B > $OUT/bench_ablate3_stamps.json 2> $OUT/bench_ablate3_stamps.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s4/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
