#!/bin/bash
# round 3, GPU session 33: one list for the whole call, a fixed number of detect workgroups sharing it out (how many?)
OUT=gpurun_out/r03_s33; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
$B --config 3 --steps 200 > $OUT/cfg3.json 2> $OUT/cfg3.err
$B --config 5 --steps 100 > $OUT/cfg5.json 2> $OUT/cfg5.err
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/cfg5_f64.err
for g in 64 128 512 1024; do
  SS_DET_LIST_WGS=$g $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_g$g.json 2> $OUT/cfg3_g$g.err
  SS_DET_LIST_WGS=$g $B --config 5 --steps 100 --diag-lib > $OUT/cfg5_g$g.json 2> $OUT/cfg5_g$g.err
done
# a dense band: every tile within reach of the threshold (start level 3 dB), culling on and off
for g in 128 256 512 1024; do
  SS_DET_LIST_WGS=$g $B --config 3 --steps 200 --diag-lib --start-level 3 > $OUT/cfg3_dense_g$g.json 2> $OUT/cfg3_dense_g$g.err
done
$B --config 3 --steps 200 --start-level 3 --no-cull > $OUT/cfg3_dense_nocull.json 2> $OUT/cfg3_dense_nocull.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s33/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-400:])
PY
