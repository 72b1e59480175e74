#!/bin/bash
# round 3, GPU session 36: listed tiles evaluated BEFORE the workgroup's column tile
OUT=gpurun_out/r03_s36; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
for rep in 1 2; do
$B --config 3 --steps 200 > $OUT/cfg3_r$rep.json 2> $OUT/cfg3.err
SS_ABLATE_ROLES=1 $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_abl1_r$rep.json 2> $OUT/cfg3.err
SS_ABLATE_ROLES=2 $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_abl2_r$rep.json 2> $OUT/cfg3.err
SS_ABLATE_ROLES=3 $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_abl3_r$rep.json 2> $OUT/cfg3.err
$B --config 3 --steps 200 --no-cull > $OUT/cfg3_nocull_r$rep.json 2> $OUT/cfg3.err
done
$B --config 3 --steps 200 --start-level 3 > $OUT/cfg3_dense.json 2> $OUT/cfg3.err
$B --config 3 --steps 200 --start-level 3 --no-cull > $OUT/cfg3_dense_nocull.json 2> $OUT/cfg3.err
$B --config 5 --steps 100 > $OUT/cfg5.json 2> $OUT/cfg5.err
SS_ABLATE_ROLES=3 $B --config 5 --steps 100 --diag-lib > $OUT/cfg5_abl3.json 2> $OUT/cfg5.err
$B --config 5 --steps 100 --no-cull > $OUT/cfg5_nocull.json 2> $OUT/cfg5.err
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/cfg5.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s36/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
