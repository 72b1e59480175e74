#!/bin/bash
# round 6, session 9: (a) does a large torch working set (as bench.py holds it) slow ss_process's pageable copy down? (b) the averager ring
# with 1024 rows for the long transforms: 262144 x 32 / x 16, config 5, one launch per call against two; the long-transform tests
OUT=gpurun_out/r06_s9
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python scripts/drop_in_trace.py --after-working-set > $OUT/drop_in_ws.txt 2>&1; cat $OUT/drop_in_ws.txt
run() { timeout 300 python bench.py --gpus 1 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub ${@:2} > $OUT/$1.json 2>/dev/null; }
for i in 1 2; do
  run x256_f32_$i --config 5 --fft 262144 --frames 32 --steps 60
  run x256_f16_$i --config 5 --fft 262144 --frames 16 --steps 80
  run x256_f64_$i --config 5 --fft 262144 --frames 64 --steps 40
  run cfg5_$i --config 5 --steps 60
  run cfg5_f64_$i --config 5 --frames 64 --steps 30
  run cfg3_$i --config 3 --steps 100
  run cf32_65536_$i --config 3 --fmt cf32 --steps 100
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s9/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config']['calls'])
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/pytest_long.txt 2>&1; tail -3 $OUT/pytest_long.txt
