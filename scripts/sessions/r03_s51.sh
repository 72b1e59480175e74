#!/bin/bash
# round 3, GPU session 51: six step-A twiddle loads per thread in the column tiles instead of fifteen — suite, then A/B on configs 3 and 5
OUT=gpurun_out/r03_s51; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
for rep in 1 2; do
for v in base colstw15; do
  $B --config 3 --steps 200 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg3_${v}_r$rep.json 2> $OUT/err
  $B --config 5 --steps 100 --lib scripts/ab/libspecscan_$v.so > $OUT/cfg5_${v}_r$rep.json 2> $OUT/err
done
done
$B --config 5 --steps 40 --frames 64 > $OUT/cfg5_f64.json 2> $OUT/err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s51/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
