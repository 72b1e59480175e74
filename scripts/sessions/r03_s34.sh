#!/bin/bash
# round 3, GPU session 34: the column workgroups take the listed tiles after their own tile; the rows kernel zeroes the list count
OUT=gpurun_out/r03_s34; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
$B --config 3 --steps 200 > $OUT/cfg3.json 2> $OUT/cfg3.err
$B --config 3 --steps 200 --no-cull > $OUT/cfg3_nocull.json 2> $OUT/cfg3_nocull.err
$B --config 5 --steps 100 > $OUT/cfg5.json 2> $OUT/cfg5.err
$B --config 5 --steps 100 --no-cull > $OUT/cfg5_nocull.json 2> $OUT/cfg5_nocull.err
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/cfg5_f64.err
$B --config 3 --steps 200 --start-level 3 > $OUT/cfg3_dense.json 2> $OUT/cfg3_dense.err
$B --config 3 --steps 200 --start-level 3 --no-cull > $OUT/cfg3_dense_nocull.json 2> $OUT/cfg3_dense_nocull.err
$B --config 3 --steps 200 --fmt cf32 > $OUT/cfg3_cf32.json 2> $OUT/cfg3_cf32.err
for o in "E*|F*,D*" "F*|E*,D*"; do
  SS_STEP_ORDER="$o" $B --config 3 --steps 200 --diag-lib > "$OUT/cfg3_order_${o//[^A-Z]/}.json" 2> /dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s34/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
