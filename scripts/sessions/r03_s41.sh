#!/bin/bash
# round 3, GPU session 41: the evaluating workgroup tests its own pair of tiles (no plan launch) against k_plan_long's list
OUT=gpurun_out/r03_s41; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
SS_TEST_USE_DIAG_LIB=1 SS_SELF_PLAN=0 timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_stated_configs.py -q -m gpu -x > $OUT/pytest_gpu_list.txt 2>&1; tail -3 $OUT/pytest_gpu_list.txt
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
for rep in 1 2; do
$B --config 3 --steps 200 --diag-lib > $OUT/cfg3_self_r$rep.json 2> $OUT/cfg3.err
SS_SELF_PLAN=0 $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_list_r$rep.json 2> $OUT/cfg3.err
$B --config 5 --steps 100 --diag-lib > $OUT/cfg5_self_r$rep.json 2> $OUT/cfg5.err
SS_SELF_PLAN=0 $B --config 5 --steps 100 --diag-lib > $OUT/cfg5_list_r$rep.json 2> $OUT/cfg5.err
done
$B --config 3 --steps 200 --no-cull > $OUT/cfg3_nocull.json 2> $OUT/cfg3.err
$B --config 5 --steps 100 --no-cull > $OUT/cfg5_nocull.json 2> $OUT/cfg5.err
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/cfg5.err
$B --config 3 --steps 200 --start-level 3 > $OUT/cfg3_dense.json 2> $OUT/cfg3.err
$B --config 3 --steps 200 --start-level 3 --no-cull > $OUT/cfg3_dense_nocull.json 2> $OUT/cfg3.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s41/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
