#!/bin/bash
# round 3, GPU session 44: 8192 points, deep pipelining — the ring rows are written by the drain, not by three frame tiles of every call
OUT=gpurun_out/r03_s44; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for rep in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/bench_default_r$rep.json 2> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 > $OUT/bench_k20_r$rep.json 2> $OUT/bench.err
done
SS_CULL_STATS=1 timeout 300 python bench.py --no-cpu-baseline --no-also --diag-lib > $OUT/bench_diag_stats.json 2> $OUT/bench_diag_stats.err; grep "specscan diag" $OUT/bench_diag_stats.err
SS_HINT_MODE=1 timeout 300 python bench.py --no-cpu-baseline --no-also --diag-lib > $OUT/bench_hint1.json 2> $OUT/bench.err
SS_ABLATE_ROLES=1 timeout 300 python bench.py --no-cpu-baseline --no-also --diag-lib > $OUT/bench_abl1.json 2> $OUT/bench.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s44/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['roofline']['launches_in_flight'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
