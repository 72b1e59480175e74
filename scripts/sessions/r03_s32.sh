#!/bin/bash
# round 3, GPU session 32: what the roles of the long transforms' step launch cost (ablations), learning over 100 frames at 2^20 points
OUT=gpurun_out/r03_s32; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
for a in 0 1 2 3; do
  SS_ABLATE_ROLES=$a $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_abl$a.json 2> $OUT/cfg3_abl$a.err
  SS_ABLATE_ROLES=$a $B --config 5 --steps 100 --diag-lib > $OUT/cfg5_abl$a.json 2> $OUT/cfg5_abl$a.err
done
$B --config 5 --steps 100 > $OUT/cfg5.json 2> $OUT/cfg5.err
$B --config 5 --steps 100 --no-cull > $OUT/cfg5_nocull.json 2> $OUT/cfg5_nocull.err
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/cfg5_f64.err
SS_STEP_ORDER="E*|F*,D*" $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_orderEFD.json 2> $OUT/cfg3_orderEFD.err
SS_STEP_ORDER="F*|D*,E*" $B --config 3 --steps 200 --diag-lib > $OUT/cfg3_orderFDE.json 2> $OUT/cfg3_orderFDE.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s32/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-400:])
PY
