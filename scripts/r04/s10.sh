#!/bin/bash
# round 4, session 10: two processes on one GPU with the wait limit at 128 polls (three runs), the single-process line for comparison,
# the wait-bound and step-pipeline tests again
OUT=gpurun_out/r04_s10
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wait_bound.py tests/test_gpu_step_pipeline.py tests/test_gpu_cull.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 2 --steps 200 --warmup 5 --no-cpu-baseline > $OUT/bench_gpus2_bands_$i.json 2> $OUT/bench_gpus2_bands_$i.err
done
timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-also --no-parity > $OUT/k200.json 2>> $OUT/ab.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity > $OUT/k20.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s10/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j.get('n_gpus'), (j.get('roofline_chain') or {}).get('frac'), j['roofline'].get('kernel_us'), j['config'].get('tiles'), j['config'].get('tail_us'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -3 $OUT/pytest.log | cut -c1-300
