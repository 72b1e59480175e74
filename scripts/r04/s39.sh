#!/bin/bash
# round 4, session 39: the final binaries (the row-tile role is known to KIND 7 only: the other instantiations of k_scan_step are back to
# their register allocation) — the whole GPU suite, smoke, the default line in the driver's form without its `also` entries, config 3
OUT=gpurun_out/r04_s39
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > $OUT/bench_k20.json 2> $OUT/bench_k20.err
timeout 600 python bench.py --no-also --no-cpu-baseline --no-live-pmc > $OUT/bench_k200.json 2> $OUT/bench_k200.err
timeout 300 python bench.py --config 3 --gpus 1 --no-cpu-baseline --no-also --no-parity --warmup 5 --steps 100 > $OUT/c3.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in ['bench_k20', 'bench_k200', 'c3']:
    try:
        j = json.loads(open(f'gpurun_out/r04_s39/{f}.json').read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'], j['roofline']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['roofline'].get('traffic_over_algorithmic'))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat $OUT/rc.txt; tail -3 $OUT/pytest_gpu.txt | cut -c1-300; tail -1 $OUT/smoke.txt | cut -c1-100
