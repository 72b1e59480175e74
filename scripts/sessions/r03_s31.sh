#!/bin/bash
# round 3, GPU session 31: tile culling for long transforms with the sliding-mean bound and the per-workgroup lists; the same bound
# in the 8192-point plan; what the rows kernel pays for the maxima and for the ring rows (A/B builds)
OUT=gpurun_out/r03_s31; mkdir -p $OUT
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
$B --config 5 --steps 40 --frames 64 --no-cull > $OUT/cfg5_f64_nocull.json 2> $OUT/cfg5_f64_nocull.err
for v in rowsnomax rowsnoring; do
  $B --config 3 --steps 200 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg3_$v.json 2> $OUT/cfg3_$v.err
  $B --config 5 --steps 100 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg5_$v.json 2> $OUT/cfg5_$v.err
done
timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
SS_CULL_STATS=1 timeout 300 python bench.py --no-cpu-baseline --no-also --diag-lib > $OUT/bench_diag_stats.json 2> $OUT/bench_diag_stats.err; grep "specscan diag" $OUT/bench_diag_stats.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s31/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f.replace('.json', '.err')).read()[-400:])
PY
