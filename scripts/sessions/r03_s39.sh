#!/bin/bash
# round 3, GPU session 39: pair p of the list to column workgroup p (no loop: the 8192-point kernel keeps its registers), one timed launch in short runs, run maxima stored contiguously
OUT=gpurun_out/r03_s39; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for rep in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/bench_default_r$rep.json 2> $OUT/bench_default.err
timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > $OUT/bench_k20_r$rep.json 2> $OUT/bench_k20.err
done
B="timeout 200 python bench.py --no-cpu-baseline --gpus 1 --warmup 5 --preheat-ms 150 --sub"
$B --config 3 --steps 200 > $OUT/cfg3.json 2> $OUT/cfg3.err
$B --config 3 --steps 200 --no-cull > $OUT/cfg3_nocull.json 2> $OUT/cfg3.err
$B --config 5 --steps 100 > $OUT/cfg5.json 2> $OUT/cfg5.err
$B --config 5 --steps 100 --no-cull > $OUT/cfg5_nocull.json 2> $OUT/cfg5.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_k20 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --preheat-ms 100 > $R/$OUT/trace_k20.log 2>&1
cd $R
cp $OUT/trace_k20/*/*_kernel_trace.csv $OUT/trace_k20.csv; rm -rf $OUT/trace_k20
python scripts/timeline_tail.py $OUT/trace_k20.csv 30 | tee $OUT/timeline_k20.txt
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s39/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['config']['candidates_per_batch'], ks)
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
