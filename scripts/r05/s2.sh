#!/bin/bash
# round 5, session 2: the radix-8 fold as the product's form of 65536-point int8 detect-mode calls (KIND 8) — first contact: the
# stated-config tests of config 3 (host path, device calls, the new steady-state case), the culling tests, then a short bench of config 3
OUT=gpurun_out/r05_s2
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -m gpu -x -q -s -k "config3" > $OUT/pytest_cfg3.txt 2>&1
tail -25 $OUT/pytest_cfg3.txt | cut -c1-400
timeout 600 python bench.py --config 3 --gpus 1 --sub --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 600 python bench.py --config 3 --frames 512 --gpus 1 --sub --no-parity --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_f512.json 2> $OUT/bench_cfg3_f512.err
python - <<'PY'
import json
for f in ['bench_cfg3.json', 'bench_cfg3_f512.json']:
    try:
        j = json.loads(open('gpurun_out/r05_s2/' + f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'], j['config']['tiles'], [(k['slot'], k['us'], k['launches_timed']) for k in j['roofline']['kernels']], (j.get('parity') or {}).get('timed_path'), (j.get('parity') or {}).get('failed'))
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r05_s2/' + f.replace('.json', '.err')).read()[-1500:])
PY
