#!/bin/bash
# round 3, GPU session 30: tile culling for long transforms (65536 / 2^20 points) — the whole GPU suite with the new culled == unculled
# cases, then configs 3 and 5 with and without culling (per-kernel event timing), 2^20 points with one-kernel rows and with
# 64-frame calls, and the default line with its `also` entries
OUT=gpurun_out/r03_s30; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -q -m gpu -s -x > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
$B --config 3 --steps 200 > $OUT/cfg3.json 2> $OUT/cfg3.err
$B --config 3 --steps 200 --no-cull > $OUT/cfg3_nocull.json 2> $OUT/cfg3_nocull.err
$B --config 5 --steps 100 > $OUT/cfg5.json 2> $OUT/cfg5.err
$B --config 5 --steps 100 --no-cull > $OUT/cfg5_nocull.json 2> $OUT/cfg5_nocull.err
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/cfg5_f64.err
SS_FFT_ROWSR=1 $B --config 5 --steps 100 --diag-lib > $OUT/cfg5_rowsr.json 2> $OUT/cfg5_rowsr.err
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s30/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
        for a in j.get('also', []):
            print('   also', a.get('baseline_config'), a.get('ms_per_step'), a.get('value'), a.get('error'), {k['slot']: k['us'] for k in a.get('kernels', [])})
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-400:])
PY
