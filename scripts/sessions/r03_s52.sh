#!/bin/bash
# round 3, GPU session 52 (final library): configs 3 and 5 culled / unculled, 64-frame calls, rocprofv3 kernel statistics of both, the driver's forms of the bench line
OUT=gpurun_out/r03_s52; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 120 python bench.py --no-cpu-baseline --config 3 --gpus 1 > $OUT/bench_cfg3.json 2> $OUT/err
timeout 120 python bench.py --no-cpu-baseline --config 3 --gpus 1 --no-cull > $OUT/bench_cfg3_nocull.json 2> $OUT/err
timeout 120 python bench.py --no-cpu-baseline --config 5 --gpus 1 > $OUT/bench_cfg5.json 2> $OUT/err
timeout 120 python bench.py --no-cpu-baseline --config 5 --gpus 1 --no-cull > $OUT/bench_cfg5_nocull.json 2> $OUT/err
timeout 120 python bench.py --no-cpu-baseline --config 5 --gpus 1 --frames 64 --steps 50 > $OUT/bench_cfg5_f64.json 2> $OUT/err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof3 -- python $R/bench.py --config 3 --gpus 1 --sub --steps 200 --warmup 10 --no-cpu-baseline > $R/$OUT/prof3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof5 -- python $R/bench.py --config 5 --gpus 1 --sub --steps 100 --warmup 10 --no-cpu-baseline > $R/$OUT/prof5.log 2>&1
cd $R
cp $OUT/prof3/*/*_kernel_stats.csv $OUT/kernel_stats_cfg3.csv; cp $OUT/prof5/*/*_kernel_stats.csv $OUT/kernel_stats_cfg5.csv; rm -rf $OUT/prof3 $OUT/prof5
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s52/bench_*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        ks = {k['slot']: k['us'] for k in j['roofline'].get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], ks)
        for a in j.get('also', []):
            print('   also', a.get('baseline_config'), a.get('frames_per_batch'), a.get('ms_per_step'), a.get('value'), a.get('error'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
head -4 $OUT/kernel_stats_cfg3.csv | cut -c1-150; head -5 $OUT/kernel_stats_cfg5.csv | cut -c1-150
